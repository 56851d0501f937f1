#!/usr/bin/env python3
"""bench.py -- Msamples/s through the FIR + demod chain on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg2|cfg1|cfg3|cfg4]

Default workload = BASELINE.json configs[1] (cfg2): one synthetic 2.4 MSPS I/Q stream,
64-tap Filter(LPF) on I and Q -> decimate by 10 -> phase-difference discriminator, fp64.
One "step" = one pass of the fused chain over a 2^28-sample block of the stream, already
resident in HBM (interleaved complex fp64 = 4 GiB), Filter/discriminator state carried from
step to step.  N > 1: one process per GPU (torch.distributed over RCCL for the barrier and
the max-over-ranks only), every rank runs its own independent stream -- weak scaling, no
data-path collective.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from sdr_amd import _lib, device, shard  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
F64_PEAK_TFLOPS = 78.6  # dense fp64 MFMA peak = fp64 vector FMA peak on MI355X (AMD datasheet: 256 CUs x 4 SIMDs x 16 lanes x 2 flop x 2.4 GHz)


def synth_fm_stream(n, seed, dev, fs=2.4e6, dev_hz=75e3, tone=1000.0, amp=100.0):
    """FM-modulated tone quantised to the radio's integer range, interleaved complex fp64 (1, n, 2)."""
    out = torch.empty(1, n, 2, dtype=torch.float64, device=dev)
    g = torch.Generator(device=dev).manual_seed(seed)
    chunk = 1 << 24
    beta = dev_hz / tone
    for k in range(0, n, chunk):
        m = min(chunk, n - k)
        t = (torch.arange(k, k + m, device=dev, dtype=torch.float64)) / fs
        ph = beta * torch.sin(2 * np.pi * tone * t)
        noise = torch.randn(m, 2, generator=g, device=dev, dtype=torch.float64) * 4.0
        out[0, k:k + m, 0] = torch.round(amp * torch.cos(ph) + noise[:, 0]).clamp_(-127, 127)
        out[0, k:k + m, 1] = torch.round(amp * torch.sin(ph) + noise[:, 1]).clamp_(-127, 127)
    return out


def lpf_taps(T, Fs, Fx):
    taps = np.zeros(T)
    err = _lib.lib().sdrx_fir_design(_lib.LPF, 0, T, Fs, Fx, 0.0, 1 << 20, taps.ctypes.data)
    assert err == 0, err
    return taps


def cpu_baseline_cfg2(taps, D, sample_log2=28):
    """The CPU restatement of the reference chain (oracle/, kind "port"), single thread, literal
    shift-register Filter (filt.cpp:248-273) on I and on Q, decimate, libm atan2 discriminator."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _oracle as O
    L = O.lib()
    n = 1 << sample_log2
    rng = np.random.default_rng(1)
    re = rng.integers(-127, 128, n).astype(np.float64)
    im = rng.integers(-127, 128, n).astype(np.float64)
    T = taps.size
    srI, srQ, pre = np.zeros(T), np.zeros(T), np.zeros(2)
    out = np.zeros(n // D + 2)
    t0 = time.perf_counter()
    L.orc_fir_execute(taps, T, srI, re, n)
    L.orc_fir_execute(taps, T, srQ, im, n)
    m = L.orc_decimate(re, n, D)
    L.orc_decimate(im, n, D)
    L.orc_fm_discriminator(re, im, m, pre, out, O.MATH_LIBM)
    dt = time.perf_counter() - t0
    return {"value": n / dt / 1e6, "unit": "Msamples/s", "cores": 1, "kind": "port",
            "sample": f"2^{sample_log2} samples of the cfg2 chain (oracle/sdr_oracle.c, gcc -O2 -ffp-contract=off), {dt:.1f} s"}


def cpu_baseline_all_cores(taps, D, threads, sample_log2=25):
    """The same port on `threads` host threads, one independent stream each (ctypes releases the GIL):
    what SURVEY 8d calls CPU-all-cores.  Reported next to cpu_baseline, never the target."""
    import threading
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _oracle as O
    L = O.lib()
    n = 1 << sample_log2
    T = taps.size
    rng = np.random.default_rng(2)
    work = []
    for _ in range(threads):
        work.append((rng.integers(-127, 128, n).astype(np.float64), rng.integers(-127, 128, n).astype(np.float64),
                     np.zeros(T), np.zeros(T), np.zeros(2), np.zeros(n // D + 2)))

    def run(w):
        re, im, srI, srQ, pre, out = w
        L.orc_fir_execute(taps, T, srI, re, n)
        L.orc_fir_execute(taps, T, srQ, im, n)
        m = L.orc_decimate(re, n, D)
        L.orc_decimate(im, n, D)
        L.orc_fm_discriminator(re, im, m, pre, out, O.MATH_LIBM)

    ths = [threading.Thread(target=run, args=(w,)) for w in work]
    t0 = time.perf_counter()
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    dt = time.perf_counter() - t0
    return {"value": threads * n / dt / 1e6, "unit": "Msamples/s", "cores": threads, "kind": "port",
            "sample": f"{threads} independent streams of 2^{sample_log2} samples, one thread each, {dt:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--workload", default="cfg2", choices=["cfg1", "cfg2", "cfg3", "cfg4", "cfg5"])
    ap.add_argument("--log2n", type=int, default=0, help="override samples per step (log2)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"], help="f32: the fp32 twins (cfg1, cfg2, cfg4)")
    ap.add_argument("--ingest", default="fp", choices=["fp", "u8"],
                    help="u8: the input is the radio's raw uint8 {I,Q} stream, converted on load (cfg1, cfg2, cfg4)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for the barrier / max-over-ranks (gloo + SDRX_BENCH_SHARE_GPU=1: "
                         "rehearse the multi-rank path on a one-GPU box, every rank on cuda:0)")
    ap.add_argument("--shard", action="store_true",
                    help="cfg4: strong scaling -- the 1024 channels (2^20 samples each) are block-partitioned over the "
                         "ranks (sdr_amd.shard.partition) instead of every rank running its own 1024")
    ap.add_argument("--decim", type=int, default=10, help="cfg2/cfg4 only: another decimation factor (exploration; the "
                                                          "BASELINE workload is 10)")
    ap.add_argument("--taps", type=int, default=64, help="cfg2/cfg4 only: another filter length (exploration)")
    ap.add_argument("--channels", type=int, default=0, help="cfg3 only: channel count (power of two; default 2^14; "
                                                            "the total stays 2^26 samples)")
    ap.add_argument("--fast-fma", action="store_true")
    ap.add_argument("--mfma", action="store_true", help="cfg5: fp64 MFMA Toeplitz formulation (fast mode 2)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if os.environ.get("SDRX_BENCH_SHARE_GPU"):
            local_rank = 0
        torch.cuda.set_device(local_rank)
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend="gloo")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    device.init(local_rank)
    n_gpus = world if world > 1 else 1

    wl = args.workload
    cfg = {}
    scaling = "weak"
    f32 = args.dtype == "f32"
    tdt = torch.float32 if f32 else torch.float64
    u8 = args.ingest == "u8"
    if (f32 or u8) and wl not in ("cfg1", "cfg2", "cfg4"):
        raise SystemExit("--dtype f32 / --ingest u8 exist for cfg1, cfg2 and cfg4")
    in_bytes = 2.0 if u8 else (8.0 if f32 else 16.0)
    lay = "raw uint8 {I,Q}, converted on load (byte - 127)" if u8 else "interleaved complex " + ("fp32" if f32 else "fp64")

    def as_input(x):  # synthetic samples are integers in [-127, 127]: byte = x + 127
        return (x + 127.0).to(torch.uint8).contiguous() if u8 else x.to(tdt)
    if wl in ("cfg2", "cfg4"):
        D, T, Fs = args.decim, args.taps, 2.4e6
        taps = lpf_taps(T, Fs, 1.0e5)
        if wl == "cfg2":
            nchan, n = 1, 1 << (args.log2n or 28)
            x = synth_fm_stream(n, 1000 + rank, dev)
        else:
            nchan, n = 1024, 1 << (args.log2n or 18)
            if args.shard:  # rank g of G owns channels [g*C/G, (g+1)*C/G), longer streams
                lo, hi = shard.partition(1024, rank, world)
                nchan, n = hi - lo, 1 << (args.log2n or 20)
                scaling = "strong"
            x = synth_fm_stream(nchan * n, 1000 + rank, dev).view(nchan, n, 2)
        x = as_input(x)
        chain = device.FirChain(taps, nchan=nchan, decim=D, device=dev, dtype=tdt)
        out = torch.zeros(nchan, chain.n_out(n), dtype=tdt, device=dev)
        step = lambda: chain.fir_fmdisc(x, out, fast_fma=args.fast_fma)  # noqa: E731
        units = nchan * n
        bytes_per_sample = in_bytes + (4.0 if f32 else 8.0) / D
        kernel = "fir_systolic_kernel<D=10, IQ_INTERLEAVED, EPI_DISC>"
        kernel = kernel if D == 10 else "fir_* (launch_fir's choice)"
        cfg = {"workload": f"{wl}: Filter(LPF,{T},2.4e6,1e5) on I,Q -> decimate {D} -> phase-difference discriminator",
               "channels_per_gpu": nchan, "samples_per_channel_per_step": n,
               "layout": lay,
               "parallelism": f"{n_gpus} independent stream set(s), no collective"}
        if scaling == "strong":
            cfg["parallelism"] = f"1024 channels block-partitioned over {n_gpus} GPU(s), no data-path collective"
    elif wl == "cfg1":
        T, Fs = 64, 2.0e6
        taps = lpf_taps(T, Fs, 1.0e5)
        nchan, n = 1, 1 << (args.log2n or 28)
        x = as_input(synth_fm_stream(n, 1000 + rank, dev))
        chain = device.FirChain(taps, nchan=nchan, decim=1, device=dev, dtype=tdt)
        out = torch.zeros(nchan, n, dtype=torch.uint8, device=dev)
        step = lambda: chain.fir_am(x, out, fast_fma=args.fast_fma)  # noqa: E731
        units = n
        bytes_per_sample = in_bytes + 1.0
        kernel = "fir_tile_kernel<R=8, IQ_INTERLEAVED, EPI_AM>"
        cfg = {"workload": "cfg1: Filter(LPF,64,2e6,1e5) on I,Q -> AM envelope threshold (ds=1)",
               "channels_per_gpu": 1, "samples_per_channel_per_step": n,
               "layout": lay,
               "parallelism": f"{n_gpus} independent stream(s), no collective"}
    elif wl == "cfg5":
        T, Fs = 1024, 2.0e6
        taps = np.zeros(T)
        assert _lib.lib().sdrx_fir_design(_lib.LPF, 0, T, Fs, Fs / 20, 0.0, 4096, taps.ctypes.data) == 0
        nchan, n = 1024, 1 << (args.log2n or 18)
        g = torch.Generator(device=dev).manual_seed(5 + rank)
        x = torch.randint(-127, 128, (nchan, n), generator=g, device=dev).to(torch.float64)
        chain = device.FirChain(taps, nchan=nchan, decim=1, device=dev)
        out = torch.zeros(nchan, n, dtype=torch.float64, device=dev)
        mode = 2 if args.mfma else (1 if args.fast_fma else 0)
        step = lambda: chain.fir(x, out, fast_fma=mode)  # noqa: E731
        units = nchan * n
        bytes_per_sample = 16.0
        flop_per_sample = 2.0 * T
        kernel = "fir_mfma_kernel<NB=4>" if mode == 2 else "fir_tile_kernel<R=8, REAL, EPI_STORE>"
        cfg = {"workload": "cfg5: 1024-tap Filter(LPF) bank, one real rail per channel, "
                           + ("fp64 MFMA Toeplitz formulation (fma chain)", "VALU convolution (fma chain)",
                              "VALU convolution (separate multiply and add roundings, as the CPU)")[2 - mode if mode else 2],
               "channels_per_gpu": nchan, "samples_per_channel_per_step": n, "layout": "real fp64",
               "parallelism": f"{n_gpus} x {nchan} independent channels, no collective"}
    else:  # cfg3
        lc = (args.channels or (1 << 14)).bit_length() - 1
        nchan, n = 1 << lc, 1 << ((args.log2n or 26) - lc)
        g = torch.Generator(device=dev).manual_seed(7 + rank)
        x = torch.randn(nchan, n, 2, generator=g, device=dev, dtype=torch.float64) * 0.3
        bank = device.PskBank(_lib.QPSK, nchan, n, device=dev)
        out = torch.zeros(nchan, bank.n_dec, dtype=torch.uint8, device=dev)
        step = lambda: bank.demod(x, out)  # noqa: E731
        units = nchan * n
        bytes_per_sample = 16.125
        kernel = "psk_loops_kernel"
        cfg = {"workload": "cfg3: QPSK matched filter /4 -> timing loop -> carrier loop /2 -> UW ambiguity",
               "channels_per_gpu": nchan, "samples_per_channel_per_step": n, "layout": "interleaved complex fp64",
               "parallelism": f"{n_gpus} x {nchan} independent channels, no collective"}

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        step()
    ev1.record()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ev_ms = ev0.elapsed_time(ev1)
    dt = shard.max_over_ranks(dt, device=dev if args.backend == "nccl" else "cpu")

    total_units = n_gpus * units
    if scaling == "strong" and dist is not None:  # ragged partitions: sum what the ranks really processed
        tu = torch.tensor([float(units)], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(tu)
        total_units = int(tu.item())
    if rank == 0:
        launch_s = ev_ms / 1e3 / args.steps  # HIP events on the launch stream, averaged over the timed steps
        achieved = bytes_per_sample * units / launch_s / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", f"pmc_traffic_{wl}.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        roof = {"bound": "hbm", "kernel": kernel, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "algorithmic_bytes_per_sample": bytes_per_sample, "launch_ms": launch_s * 1e3}
        if wl == "cfg5":  # 2048 flop per 16 B: compute bound; priced against the dense fp64 matrix peak (= the vector FMA peak)
            tf = flop_per_sample * units / launch_s / 1e12
            roof = {"bound": "mfma", "kernel": kernel, "achieved": tf, "peak": F64_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": tf / F64_PEAK_TFLOPS, "traffic": traffic, "algorithmic_flop_per_sample": flop_per_sample,
                    "launch_ms": launch_s * 1e3}
        res = {
            "metric": "Msamples/sec through FIR+demod chain",
            "value": total_units * args.steps / dt / 1e6,
            "unit": "Msamples/s",
            "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": cfg,
            "roofline": roof,
        }
        if n_gpus == 1 and not args.no_cpu and wl == "cfg2" and not f32 and not u8:
            res["cpu_baseline"] = cpu_baseline_cfg2(taps, 10)
            threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(16, os.cpu_count() or 1)
            res["cpu_baseline_all_cores"] = cpu_baseline_all_cores(taps, 10, threads)
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
