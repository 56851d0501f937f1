#!/usr/bin/env python3
"""bench.py -- Msamples/s through the FIR + demod chain on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg1|cfg2|cfg3|cfg4|cfg5]

N = 1 (default): BASELINE.json configs[1] (cfg2): one synthetic 2.4 MSPS I/Q stream, 64-tap Filter(LPF) on I and Q ->
decimate by 10 -> phase-difference discriminator, fp64.  One "step" = one pass of the fused chain over a 2^28-sample
block of the stream, already resident in HBM (interleaved complex fp64 = 4 GiB), Filter / discriminator state carried
from step to step.  The line also carries short runs of the other BASELINE configs under "extra" (--no-extra skips).

N > 1: BASELINE.json configs[3] (cfg4): 1024 independent channels x 2^20 samples of the same chain, block-partitioned
over the ranks (sdr_amd.shard.partition: rank g owns channels [g*1024/N, (g+1)*1024/N)) -- strong scaling, no data-path
collective; torch.distributed carries only the barrier, the max-over-ranks and the unit count (over gloo as CPU
tensors by default, RCCL left for GPU tensors: --backend).
`--gpus 1 --workload cfg4` is the N = 1 point of that curve.

One process per GPU.  Under `python -m torch.distributed.run ... bench.py --gpus N` every rank runs worker().  Started
plainly as `python bench.py --gpus N` (no WORLD_SIZE in the environment) this process only SPAWNS the N rank processes
-- before it touches the GPU, without ever exec'ing -- relays rank 0's JSON line and exits with the worst exit code.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
F64_PEAK_TFLOPS = 78.6  # dense fp64 MFMA peak = fp64 vector FMA peak on MI355X (256 CUs x 4 SIMDs x 16 lanes x 2 flop x 2.4 GHz)
F32_PEAK_TFLOPS = 157.3  # fp32 vector (packed FMA) peak
I8_PEAK_TOPS = 5000.0  # dense int8 matrix peak (v_mfma_i32_16x16x64_i8: the guide's dense fp8 figure)
F64_VALU_2R_TFLOPS = 39.3  # two-rounding arithmetic (separate v_mul_f64 + v_add_f64): one flop per lane and cycle


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--workload", default=None, choices=["cfg1", "cfg2", "cfg3", "cfg4", "cfg5"],
                    help="default: cfg2 on one GPU, cfg4 (1024 channels sharded over the ranks) on several")
    ap.add_argument("--log2n", type=int, default=0, help="override samples per channel per step (log2)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline legs")
    ap.add_argument("--no-extra", action="store_true", help="skip the short runs of the other configs (extra.configs)")
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"], help="f32: the fp32 twins (cfg1, cfg2, cfg4)")
    ap.add_argument("--ingest", default="fp", choices=["fp", "u8"],
                    help="u8: the input is the radio's raw uint8 {I,Q} stream, converted on load (cfg1, cfg2, cfg4)")
    ap.add_argument("--backend", default="auto", choices=["auto", "nccl", "gloo"],
                    help="torch.distributed backend.  The data path has no collective: the only traffic between ranks is "
                         "the timing barrier and two scalars.  auto (default): process group 'cpu:gloo,cuda:nccl' -- those "
                         "go over gloo as CPU tensors, RCCL stays available for GPU tensors (shard.gather_bits) and is not "
                         "initialised unless used; nccl: barrier and scalars as GPU tensors over RCCL; gloo (+ "
                         "SDRX_BENCH_SHARE_GPU=1): rehearse the multi-rank path on a one-GPU box, every rank on cuda:0")
    ap.add_argument("--rccl-selftest", action="store_true",
                    help="N = 1 only: form a one-rank process group and run the N > 1 epilogue (shard.gather_bits over RCCL) "
                         "after the timed region, so that a one-GPU box shows the RCCL path executing")
    ap.add_argument("--shard", action="store_true", help="(kept for older command lines: cfg4 is always sharded)")
    ap.add_argument("--weak", action="store_true",
                    help="cfg4: every rank runs its own 1024 channels x 2^18 instead of a share of 1024 x 2^20")
    ap.add_argument("--decim", type=int, default=10, help="cfg2/cfg4 only: another decimation factor (exploration)")
    ap.add_argument("--taps", type=int, default=64, help="cfg2/cfg4 only: another filter length (exploration)")
    ap.add_argument("--channels", type=int, default=0, help="cfg3: channel count (power of two; default 2^14; the total "
                                                            "stays 2^26 samples); cfg4: total channels (default 1024)")
    ap.add_argument("--bpsk", action="store_true", help="cfg3: BPSK instead of QPSK")
    ap.add_argument("--fast-fma", action="store_true")
    ap.add_argument("--spin-up-ms", type=float, default=250.0,
                    help="untimed launches of the same step for this long BEFORE the W warm-up steps: the part's clocks take "
                         "tens of milliseconds of sustained load to reach the state they hold in a running pipeline, and fall "
                         "back within half a second of idling (profiles/probes/warm_order_probe.py).  0 = none.  The line "
                         "reports the figure without it too (spin_up.cold_ms_per_step)")
    ap.add_argument("--mfma", action="store_true", help="cfg5: fp64 MFMA Toeplitz formulation (fast mode 2): cfg5's default")
    ap.add_argument("--exact", action="store_true",
                    help="cfg5: the VALU convolution with the CPU's two roundings per tap (bit-identical samples) instead of the "
                         "MFMA formulation (fused chain, within north_star's 1e-12 on filtered samples)")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------ N > 1 without a launcher

def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(n, argv):
    """Start n fresh rank processes (this process has not touched the GPU and never execs), relay rank 0's stdout,
    return the worst exit code.  A rank that fails takes the others down (they would wait in a barrier for ever)."""
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
        out = subprocess.PIPE if r == 0 else sys.stderr  # one JSON line on stdout: rank 0's
        procs.append(subprocess.Popen([sys.executable, "-u", os.path.abspath(__file__)] + list(argv), env=env, stdout=out))
    worst = 0
    try:
        for line in procs[0].stdout:  # ends when rank 0 closes its stdout
            sys.stdout.write(line.decode(errors="replace"))
            sys.stdout.flush()
        pending = set(range(n))
        while pending:
            for r in sorted(pending):
                rc = procs[r].poll()
                if rc is None:
                    continue
                pending.discard(r)
                if rc != 0:
                    worst = worst or (rc if rc > 0 else 1)
                    print(f"bench.py: rank {r} exited with {rc}", file=sys.stderr)
                    for q in pending:  # exactly the processes started above
                        procs[q].kill()
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return worst


# ------------------------------------------------------------------------------------------ workloads

def synth_fm_stream(torch, n, seed, dev, fs=2.4e6, dev_hz=75e3, tone=1000.0, amp=100.0, restart=0):
    """FM-modulated tone quantised to the radio's integer range, interleaved complex fp64 (1, n, 2).  restart > 0:
    the time axis restarts every `restart` samples (independent channels laid end to end)."""
    import numpy as np
    out = torch.empty(1, n, 2, dtype=torch.float64, device=dev)
    g = torch.Generator(device=dev).manual_seed(seed)
    chunk = 1 << 24
    beta = dev_hz / tone
    for k in range(0, n, chunk):
        m = min(chunk, n - k)
        idx = torch.arange(k, k + m, device=dev, dtype=torch.float64)
        if restart:
            idx = idx % restart
        ph = beta * torch.sin(2 * np.pi * tone * (idx / fs))
        noise = torch.randn(m, 2, generator=g, device=dev, dtype=torch.float64) * 4.0
        out[0, k:k + m, 0] = torch.round(amp * torch.cos(ph) + noise[:, 0]).clamp_(-127, 127)
        out[0, k:k + m, 1] = torch.round(amp * torch.sin(ph) + noise[:, 1]).clamp_(-127, 127)
    return out


def synth_ook_stream(torch, n, seed, dev, amp=100.0, spb=8, noise=10.0, phase=0.7):
    """SURVEY 8d's cfg1 input: an on-off-keyed carrier (amplitude `amp` on random bits, `spb` samples per bit) + uniform
    noise of +-`noise`, quantised to the radio's integer range; interleaved complex fp64 (1, n, 2).  The filtered
    envelope crosses AM.h's threshold of 80 at nearly every bit edge."""
    import math
    out = torch.empty(1, n, 2, dtype=torch.float64, device=dev)
    g = torch.Generator(device=dev).manual_seed(seed)
    chunk = 1 << 24  # a multiple of spb
    c, s = amp * math.cos(phase), amp * math.sin(phase)
    for k in range(0, n, chunk):
        m = min(chunk, n - k)
        bits = torch.randint(0, 2, ((m + spb - 1) // spb,), generator=g, device=dev).to(torch.float64)
        env = bits.repeat_interleave(spb)[:m]
        u = (torch.rand(m, 2, generator=g, device=dev, dtype=torch.float64) * 2.0 - 1.0) * noise
        out[0, k:k + m, 0] = torch.round(env * c + u[:, 0]).clamp_(-127, 127)
        out[0, k:k + m, 1] = torch.round(env * s + u[:, 1]).clamp_(-127, 127)
    return out


def synth_psk_bank(torch, bank, kind, nchan, n, seed, dev, snr_db=20.0, uw_every=256):
    """SURVEY 8d's cfg3 input, made on the device with the library's own modulator (sdrx_psk_modulate_dev = the
    reference's encode_LUT + encode_pulseShape, 8 samples per symbol): random symbols framed with the demodulator's unique
    word every `uw_every` symbols (placed reversed, the way the ambiguity pass compares: BPSK.cpp:456, QPSK.cpp:428), carrier
    phase 0 / 90 / 180 / 270 degrees by channel, a timing offset of channel % 5 samples, white Gaussian noise at `snr_db`.
    The whole bank is modulated as one symbol stream and cut into channels.  -> (nchan, n, 2) interleaved complex fp64."""
    import ctypes as C
    import math

    from sdr_amd import _lib
    L = _lib.lib()
    nsym = n // 8
    g = torch.Generator(device=dev).manual_seed(seed)
    bpsk = kind == _lib.BPSK
    sym = torch.randint(0, 2 if bpsk else 4, (nchan, nsym), generator=g, device=dev, dtype=torch.uint8)
    st0 = _lib.PskState.from_buffer_copy(bank.states[0].cpu().numpy().tobytes())
    # the state keeps a BPSK word in uw[0][0 .. size) and a QPSK word in the LAST `size` of its 64 entries (QPSK.cpp:96-110)
    uw = [int(st0.uw[0][j if bpsk else 64 - st0.uw_size + j]) for j in range(st0.uw_size)]
    pat = torch.tensor(uw[::-1], dtype=torch.uint8, device=dev)
    for at in range(96, nsym - len(uw), uw_every):
        sym[:, at:at + len(uw)] = pat
    re = torch.empty(nchan * n, dtype=torch.float64, device=dev)
    im = re if bpsk else torch.empty_like(re)  # BPSK: both rails carry the same data (TEST_Modulator_Demodulator.cpp:41-46)
    _lib.check(L.sdrx_psk_modulate_dev(kind, C.c_void_p(sym.data_ptr()), nchan * nsym, C.c_void_p(re.data_ptr()),
                                       None if bpsk else C.c_void_p(im.data_ptr()), nchan * n,
                                       C.c_void_p(torch.cuda.current_stream().cuda_stream)), "sdrx_psk_modulate_dev")
    re, im = re.view(nchan, n), im.view(nchan, n)
    x = torch.empty(nchan, n, 2, dtype=torch.float64, device=dev)
    sigma = math.sqrt(float((re * re + im * im).mean().item()) / (10.0 ** (snr_db / 10.0)) / 2.0)
    for q, (c, s_) in enumerate(((1.0, 0.0), (0.0, 1.0), (-1.0, 0.0), (0.0, -1.0))):  # z * e^{j 90 q}
        x[q::4, :, 0] = re[q::4] * c - im[q::4] * s_
        x[q::4, :, 1] = im[q::4] * c + re[q::4] * s_
    for k in range(1, 5):
        x[k::5] = torch.roll(x[k::5], k, dims=1)
    for lo in range(0, nchan, 1024):
        x[lo:lo + 1024] += torch.randn(x[lo:lo + 1024].shape, generator=g, device=dev, dtype=torch.float64) * sigma
    return x


def lpf_taps(T, Fs, Fx, max_taps=1 << 20):
    import numpy as np
    from sdr_amd import _lib
    taps = np.zeros(T)
    err = _lib.lib().sdrx_fir_design(_lib.LPF, 0, T, Fs, Fx, 0.0, max_taps, taps.ctypes.data)
    assert err == 0, err
    return taps


def build_workload(wl, args, rank, world, dev):
    """-> dict(step, units, bytes_per_sample, flop_per_sample, cfg, scaling, taps, D): one step = one pass of the
    chain over this rank's resident block."""
    import torch
    from sdr_amd import _lib, device, shard
    f32 = args.dtype == "f32"
    tdt = torch.float32 if f32 else torch.float64
    u8 = args.ingest == "u8"
    if (f32 or u8) and wl not in ("cfg1", "cfg2", "cfg4"):
        raise SystemExit("--dtype f32 / --ingest u8 exist for cfg1, cfg2 and cfg4")
    in_bytes = 2.0 if u8 else (8.0 if f32 else 16.0)
    lay = "raw uint8 {I,Q}, converted on load (byte - 127)" if u8 else "interleaved complex " + ("fp32" if f32 else "fp64")

    def as_input(x):  # synthetic samples are integers in [-127, 127]: byte = x + 127
        return (x + 127.0).to(torch.uint8).contiguous() if u8 else x.to(tdt)

    w = {"scaling": "weak", "flop_per_sample": None, "taps": None, "D": None, "report": None}
    if wl in ("cfg2", "cfg4"):
        D, T, Fs = args.decim, args.taps, 2.4e6
        taps = lpf_taps(T, Fs, 1.0e5)
        if wl == "cfg2":
            nchan, n = 1, 1 << (args.log2n or 28)
            x = synth_fm_stream(torch, n, 1000 + rank, dev)
            par = f"{world} independent stream(s), one per GPU, no collective"
        elif args.weak:
            nchan, n = args.channels or 1024, 1 << (args.log2n or 18)
            x = synth_fm_stream(torch, nchan * n, 1000 + rank, dev, restart=n).view(nchan, n, 2)
            par = f"{world} x {nchan} independent channels, no collective"
        else:  # SURVEY 8e / BASELINE configs[3]: rank g of G owns channels [g*C/G, (g+1)*C/G)
            total = args.channels or 1024
            lo, hi = shard.partition(total, rank, world)
            nchan, n = hi - lo, 1 << (args.log2n or 20)
            w["scaling"] = "strong"
            x = synth_fm_stream(torch, max(nchan, 1) * n, 1000 + lo, dev, restart=n).view(max(nchan, 1), n, 2)[:nchan]
            par = f"{total} channels block-partitioned over {world} GPU(s), no data-path collective"
        x = as_input(x)
        chain = device.FirChain(taps, nchan=nchan, decim=D, device=dev, dtype=tdt)
        out = torch.zeros(nchan, chain.n_out(n), dtype=tdt, device=dev)
        w.update(step=lambda: chain.fir_fmdisc(x, out, fast_fma=args.fast_fma), units=nchan * n,
                 bytes_per_sample=in_bytes + (4.0 if f32 else 8.0) / D, taps=taps, D=D,
                 cfg={"workload": f"{wl}: Filter(LPF,{T},2.4e6,1e5) on I,Q -> decimate {D} -> phase-difference discriminator",
                      "channels_per_gpu": nchan, "samples_per_channel_per_step": n, "layout": lay, "parallelism": par})
    elif wl == "cfg1":
        T, Fs = 64, 2.0e6
        taps = lpf_taps(T, Fs, 1.0e5)
        nchan, n = 1, 1 << (args.log2n or 28)
        x = as_input(synth_ook_stream(torch, n, 1 + rank, dev))
        chain = device.FirChain(taps, nchan=nchan, decim=1, device=dev, dtype=tdt)
        out = torch.zeros(nchan, n, dtype=torch.uint8, device=dev)
        w.update(step=lambda: chain.fir_am(x, out, fast_fma=args.fast_fma), units=n, bytes_per_sample=in_bytes + 1.0,
                 flop_per_sample=4.0 * T,  # 2 rails x (mul + add) per tap
                 cfg={"workload": "cfg1: Filter(LPF,64,2e6,1e5) on I,Q -> AM envelope threshold (ds=1)",
                      "channels_per_gpu": 1, "samples_per_channel_per_step": n, "layout": lay,
                      "parallelism": f"{world} independent stream(s), no collective"})
    elif wl == "cfg5":
        T, Fs = 1024, 2.0e6
        taps = lpf_taps(T, Fs, Fs / 20, 4096)
        nchan, n = 1024, 1 << (args.log2n or 18)
        g = torch.Generator(device=dev).manual_seed(5 + rank)
        x = torch.randint(-127, 128, (nchan, n), generator=g, device=dev).to(torch.float64)
        chain = device.FirChain(taps, nchan=nchan, decim=1, device=dev)
        out = torch.zeros(nchan, n, dtype=torch.float64, device=dev)
        mode = 0 if args.exact else (1 if args.fast_fma else 2)  # cfg5 hands back filtered SAMPLES: the 1e-12 tolerance is north_star's own
        w.update(step=lambda: chain.fir(x, out, fast_fma=mode), units=nchan * n, bytes_per_sample=16.0,
                 flop_per_sample=2.0 * T,
                 cfg={"workload": "cfg5: 1024-tap Filter(LPF) bank, one real rail per channel, "
                                  + ("VALU convolution (separate multiply and add roundings, as the CPU)",
                                     "VALU convolution (fma chain)", "fp64 MFMA Toeplitz formulation (fma chain)")[mode],
                      "channels_per_gpu": nchan, "samples_per_channel_per_step": n, "layout": "real fp64",
                      "parallelism": f"{world} x {nchan} independent channels, no collective"})
    else:  # cfg3
        lc = (args.channels or (1 << 14)).bit_length() - 1
        nchan, n = 1 << lc, 1 << ((args.log2n or 26) - lc)
        kind = _lib.BPSK if args.bpsk else _lib.QPSK
        bank = device.PskBank(kind, nchan, n, device=dev)
        x = synth_psk_bank(torch, bank, kind, nchan, n, 7 + rank, dev)
        out = torch.zeros(nchan, bank.n_dec, dtype=torch.uint8, device=dev)
        hits_at = _lib.PskState.n_hits.offset

        def report():  # the unique-word hits the last step's ambiguity pass found: the fold ran on real frames
            h = bank.states[:, hits_at:hits_at + 4].contiguous().view(torch.int32).view(-1)
            return {"uw_hits_last_step": int(h.sum().item()), "channels_with_uw_hits": int((h > 0).sum().item()),
                    "input": "framed symbols (unique word every 256), product modulator, phase 0/90/180/270 by channel, "
                             "timing offset channel % 5, AWGN 20 dB"}
        w["report"] = report
        w.update(step=lambda: bank.demod(x, out), units=nchan * n, bytes_per_sample=16.125,
                 flop_per_sample=129.0 * 2 * 2 / 4,  # matched filter only: 129 taps x 2 rails x (mul + add), every 4th output
                 cfg={"workload": ("cfg3: BPSK" if args.bpsk else "cfg3: QPSK")
                                  + " matched filter /4 -> timing loop -> carrier loop /2 -> UW ambiguity",
                      "channels_per_gpu": nchan, "samples_per_channel_per_step": n, "layout": "interleaved complex fp64",
                      "parallelism": f"{world} x {nchan} independent channels, no collective"})
    return w


def timed_steps(torch, step, warmup, steps, barrier=None, spin_up_s=0.0):
    """-> (wall seconds between the barriers, HIP-event milliseconds on the launch stream) for `steps` steps."""
    if spin_up_s > 0:  # bring the device to its sustained state (same launches, untimed), then the warm-up steps as asked
        t_end = time.perf_counter() + spin_up_s
        while time.perf_counter() < t_end:
            for _ in range(8):
                step()
            torch.cuda.synchronize()
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    if barrier is not None:
        barrier()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()  # the kernels are launched on torch's current stream (device._stream()), the one these events see
    for _ in range(steps):
        step()
    ev1.record()
    torch.cuda.synchronize()
    if barrier is not None:
        barrier()
    torch.cuda.synchronize()
    return time.perf_counter() - t0, ev0.elapsed_time(ev1)


def roofline_of(wl, w, launch_s, kernel):
    achieved = w["bytes_per_sample"] * w["units"] / launch_s / 1e9
    traffic, src = None, None
    tpath = os.path.join(ROOT, "profiles", f"pmc_traffic_{wl}.json")
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get("hbm_bytes_per_launch")
            src = f"profiles/pmc_traffic_{wl}.json (rocprofv3 --pmc, a separate profiled run of this workload; not measured in this run)"
        except Exception:
            traffic = None
    roof = {"bound": "hbm", "kernel": kernel, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": src,
            "algorithmic_bytes_per_sample": w["bytes_per_sample"], "launch_ms": launch_s * 1e3}
    if w["flop_per_sample"]:
        tf = w["flop_per_sample"] * w["units"] / launch_s / 1e12
        fma = "fma" in kernel or "mfma" in kernel
        amv = "fir_amv" in kernel
        i8 = amv and "int8" in kernel
        peak = I8_PEAK_TOPS if i8 else (F32_PEAK_TFLOPS if amv else (F64_PEAK_TFLOPS if fma else F64_VALU_2R_TFLOPS))
        comp = {"achieved": tf, "peak": peak, "unit": "TFLOP/s", "frac": tf / peak,
                "peak_is": ("int8 matrix-core dense peak; achieved counts the filter's ALGORITHMIC multiply-adds -- the kernel executes six times "
                            "as many (three 7-bit digit planes of the taps x 128 Toeplitz columns for 64 taps) and is not bound by them: "
                            "samples as 8-bit integers, decisions verified / recomputed in fp64 (bits of the exact path)") if i8
                else ("fp32 packed-FMA vector peak (fp32 filter, decisions verified / recomputed in fp64: bits of the exact path)" if amv
                else ("fp64 FMA / MFMA peak" if fma else "fp64 VALU peak of two-rounding arithmetic (mul and add issue separately)")),
                "algorithmic_flop_per_sample": w["flop_per_sample"]}
        if wl == "cfg5":  # 2048 flop per 16 B: compute bound; priced against the dense fp64 matrix peak (= the vector FMA peak)
            roof = {"bound": "mfma", "kernel": kernel, "achieved": tf, "peak": F64_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": tf / F64_PEAK_TFLOPS, "traffic": traffic, "traffic_source": src,
                    "algorithmic_flop_per_sample": w["flop_per_sample"], "launch_ms": launch_s * 1e3}
        else:
            roof["compute"] = comp
    return roof


# ------------------------------------------------------------------------------------------ CPU baseline (oracle/, port)

def cpu_baseline_cfg2(taps, D, sample_log2=28):
    """The CPU restatement of the reference chain (oracle/, kind "port"), single thread, literal
    shift-register Filter (filt.cpp:248-273) on I and on Q, decimate, libm atan2 discriminator."""
    import numpy as np
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
            "sample": f"2^{sample_log2} samples of the cfg2 chain (T={T}, D={D}; oracle/sdr_oracle.c, gcc -O2 -ffp-contract=off), {dt:.1f} s"}


def cpu_baseline_all_cores(taps, D, threads, sample_log2=25):
    """The same port on `threads` host threads, one independent stream each (ctypes releases the GIL):
    what SURVEY 8d calls CPU-all-cores.  Reported next to cpu_baseline, never the target."""
    import threading

    import numpy as np
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

    def run(wk):
        re, im, srI, srQ, pre, out = wk
        L.orc_fir_execute(taps, T, srI, re, n)
        L.orc_fir_execute(taps, T, srQ, im, n)
        m = L.orc_decimate(re, n, D)
        L.orc_decimate(im, n, D)
        L.orc_fm_discriminator(re, im, m, pre, out, O.MATH_LIBM)

    ths = [threading.Thread(target=run, args=(wk,)) for wk in work]
    t0 = time.perf_counter()
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    dt = time.perf_counter() - t0
    return {"value": threads * n / dt / 1e6, "unit": "Msamples/s", "cores": threads, "kind": "port",
            "sample": f"{threads} independent streams of 2^{sample_log2} samples (T={T}, D={D}), one thread each, {dt:.1f} s"}


def host_threads():
    """The CPU share this process may use: OMP_NUM_THREADS where the box sets it (16 per GPU there), else the affinity mask."""
    v = int(os.environ.get("OMP_NUM_THREADS", "0") or 0)
    if v > 0:
        return v
    try:
        return len(os.sched_getaffinity(0))
    except AttributeError:
        return os.cpu_count() or 1


# ------------------------------------------------------------------------------------------ RCCL epilogue (N > 1)

def gpu_identity(torch, local_rank):
    """(host, PCI domain, bus, device) of this rank's GPU as 4 int64: ranks that share a GPU cannot form an RCCL communicator."""
    import zlib
    pr = torch.cuda.get_device_properties(local_rank)
    return [zlib.crc32(socket.gethostname().encode()), int(getattr(pr, "pci_domain_id", 0)), int(getattr(pr, "pci_bus_id", -1)),
            int(getattr(pr, "pci_device_id", -1))]


def rccl_gather_epilogue(torch, dist, shard, device, args, rank, world, dev, red_dev, out):
    """SURVEY 8e / north_star's "RCCL over xGMI only for an optional final gather", OUTSIDE the timed region: every rank
    demodulates a small block of its own channels (AM threshold on the real rail of 2^14 samples per channel, ds = 8:
    2048 decision bytes per channel), the rows are collected on rank 0 with shard.gather_bits over RCCL (GPU tensors), and
    rank 0 checks every gathered row against the checksum its owner sent over the control backend.  Fills `out`."""
    total = args.channels or 1024
    lo, hi = shard.partition(total, rank, world)
    n, ds = 1 << 14, 8
    x = synth_fm_stream(torch, max(hi - lo, 1) * n, 5000 + lo, dev, restart=n).view(max(hi - lo, 1), n, 2)[: hi - lo]
    re = x[:, :, 0].contiguous()
    bits = torch.zeros(hi - lo, n // ds, dtype=torch.uint8, device=dev)
    if hi > lo:
        device.am_demod(re, torch.zeros_like(re), bits, ds)
    torch.cuda.synchronize()
    cs_all = shard.gather_checksums(shard.row_checksums(bits), total, device=red_dev)
    times = []
    full = None
    for _ in range(3):  # the first call brings the communicator up
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        full = shard.gather_bits(bits, total, dst=0, always=True)
        torch.cuda.synchronize()
        times.append((time.perf_counter() - t0) * 1e3)
    out.update({"rccl_ranks": world, "rccl_init_and_first_gather_ms": times[0], "gather_ms": min(times[1:]),
                "gather_bytes": total * (n // ds), "gather_rows": total})
    if rank == 0:
        ok = full is not None and full.shape[0] == total and bool(torch.equal(shard.row_checksums(full), cs_all))
        out["gather_verified"] = ok
        out["gather_ones_fraction"] = float(full.float().mean().item()) if full is not None and full.numel() else None


# ------------------------------------------------------------------------------------------ one rank

def worker(args):
    # stdout carries ONE line, the JSON: everything else a library prints there (gloo announces its connections on stdout)
    # goes to stderr.  The descriptor itself is switched, so native code is covered; the line is written to the saved one.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    import torch
    sys.path.insert(0, ROOT)
    from sdr_amd import _lib, device, shard

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1 or args.rccl_selftest:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            os.environ.setdefault("MASTER_PORT", str(_free_port()))
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if os.environ.get("SDRX_BENCH_SHARE_GPU"):
            local_rank = 0
        ndev = torch.cuda.device_count()
        if ndev and local_rank >= ndev:  # a launcher that shows every rank only its own GPU (HIP_VISIBLE_DEVICES per rank)
            local_rank %= ndev
        torch.cuda.set_device(local_rank)
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        elif args.backend == "gloo":
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="cpu:gloo,cuda:nccl")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    device.init(local_rank)
    # host threads of a rank next to its GPU (launches, pinned staging): only where several ranks share the host
    numa_node = shard.pin_to_gpu_numa_node(local_rank) if world > 1 and not os.environ.get("SDRX_BENCH_SHARE_GPU") else None
    red_dev = dev if args.backend == "nccl" else "cpu"
    L = _lib.lib()

    wl = args.workload or ("cfg2" if world == 1 else "cfg4")
    w = build_workload(wl, args, rank, world, dev)
    barrier = None
    if dist is not None:  # (an all-reduce of one element on the reduction device: with the mixed backend dist.barrier()
        barrier = lambda: dist.all_reduce(torch.zeros(1, device=red_dev))  # noqa: E731   would bring up RCCL for nothing)
    cold_ms = None
    if world == 1 and args.spin_up_ms > 0:  # the same measurement first thing, for the record (not the value)
        cold_ms = timed_steps(torch, w["step"], args.warmup, args.steps)[1] / args.steps
    dt, ev_ms = timed_steps(torch, w["step"], args.warmup, args.steps, barrier, args.spin_up_ms / 1e3)
    kernel = L.sdrx_last_kernel().decode()  # what the launcher picked for this shape (the last step's launches)
    dt = shard.max_over_ranks(dt, device=red_dev)

    total_units = world * w["units"]
    rccl = {}
    if dist is not None:  # ragged partitions: sum what the ranks really processed
        tu = torch.tensor([float(w["units"])], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tu)
        total_units = int(tu.item())
        # ---- after the timed region: the optional final gather over RCCL, when every rank has a GPU of its own ----
        ident = torch.tensor(gpu_identity(torch, local_rank), dtype=torch.int64, device=red_dev)
        ids = [torch.empty_like(ident) for _ in range(world)]
        dist.all_gather(ids, ident)
        distinct = len({tuple(int(v) for v in t.tolist()) for t in ids}) == world
        if args.backend == "gloo" or not distinct:
            rccl = {"rccl_ranks": 0, "rccl_skipped": "backend gloo" if args.backend == "gloo" else "ranks share a GPU"}
        else:
            import threading

            def run():
                try:
                    torch.cuda.set_device(local_rank)
                    rccl_gather_epilogue(torch, dist, shard, device, args, rank, world, dev, red_dev, rccl)
                except Exception as ex:  # the epilogue must never cost the headline line
                    rccl.update({"rccl_ranks": 0, "rccl_error": repr(ex)[:300]})

            th = threading.Thread(target=run, daemon=True)
            th.start()
            th.join(float(os.environ.get("SDRX_BENCH_RCCL_TIMEOUT_S", "120")))
            if th.is_alive():  # a communicator that never comes up: report it, print the line, leave without the final barrier
                rccl = {"rccl_ranks": 0, "rccl_error": "timeout"}
    if rank == 0:
        launch_s = ev_ms / 1e3 / args.steps  # HIP events on the launch stream, averaged over the timed steps
        res = {
            "metric": "Msamples/sec through FIR+demod chain",
            "value": total_units * args.steps / dt / 1e6,
            "unit": "Msamples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": w["scaling"], "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": w["cfg"],
            "roofline": roofline_of(wl, w, launch_s, kernel),
        }
        res["spin_up"] = {"ms": args.spin_up_ms, "cold_ms_per_step": cold_ms,
                          "what": "untimed launches of the same step before the W warm-up steps (device at the clocks it holds in a "
                                  "running pipeline); cold_ms_per_step = the same W + K measurement first thing in the process"}
        if dist is not None:
            res.update({k: v for k, v in rccl.items() if k in ("rccl_ranks", "gather_ms")})
            res["rccl"] = dict(rccl, numa_node=numa_node, what="epilogue after the timed region: shard.gather_bits over RCCL, "
                               "rows verified on rank 0 against checksums sent over the control backend; the timed region has no collective")
        if w["report"]:
            res["config"] = dict(res["config"], **w["report"]())
        plain = args.dtype == "f64" and args.ingest == "fp"
        taps, D = w["taps"], w["D"]
        del w
        torch.cuda.empty_cache()
        if world == 1 and not args.no_extra and wl == "cfg2" and plain:
            res["extra"] = {"configs": extra_configs(torch, args, dev, L), "host_layer": host_layer(torch, dev, L),
                            "note": "short runs (20 warm-up + 50 timed steps) of the other BASELINE configs in this process, "
                                    "after the headline; frac_hbm = algorithmic bytes / launch time / 8 TB/s"}
        if world == 1 and not args.no_cpu and wl == "cfg2" and plain:
            res["cpu_baseline"] = cpu_baseline_cfg2(taps, D)
            threads = host_threads()
            res["cpu_baseline_all_cores"] = cpu_baseline_all_cores(taps, D, threads, 25 if threads <= 16 else 23)
        os.write(json_fd, (json.dumps(res) + "\n").encode())
    if dist is not None:
        if rccl.get("rccl_error") == "timeout":
            sys.stdout.flush()
            os._exit(0)  # (a thread is still inside RCCL: tearing the group down would wait for it)
        try:
            barrier()
            dist.destroy_process_group()
        except Exception as ex:  # a peer that left after an RCCL timeout: the line is out already
            print(f"bench.py: rank {rank}: {ex!r} at the final barrier", file=sys.stderr)


def extra_configs(torch, args, dev, L):
    """cfg1 (fp64 samples, and the radio's raw bytes) / cfg3 (QPSK and BPSK) / cfg4 (the N = 1 point of the sharded curve) / cfg5 (VALU exact, MFMA), a few steps
    each, so that the driver's box times them too.  Each entry: Msamples/s, ms per launch, fractions of the roofs."""
    out = {}
    # cfg3's 2^26 samples: as 2^14 channels x 2^12 (enough channels to put a wave pair on half of the SIMDs) and as the
    # 1024 channels x 2^16 of SURVEY 8d (16 wave pairs walking 16384 serial loop steps each: latency-bound by design)
    runs = [("cfg1", []), ("cfg1_u8", ["--ingest", "u8"]), ("cfg3", []), ("cfg3_bpsk", ["--bpsk"]), ("cfg3_1024ch", ["--channels", "1024"]), ("cfg4", []),
            ("cfg5", []), ("cfg5_exact", ["--exact"])]
    for name, flags in runs:
        a = parse_args(["--workload", name.split("_")[0], "--steps", "50", "--warmup", "20"] + flags)
        try:
            w = build_workload(a.workload, a, 0, 1, dev)
            dt, ev_ms = timed_steps(torch, w["step"], a.warmup, a.steps)
            kernel = L.sdrx_last_kernel().decode()
            launch_s = ev_ms / 1e3 / a.steps
            roof = roofline_of(a.workload, w, launch_s, kernel)
            e = {"workload": w["cfg"]["workload"], "channels": w["cfg"]["channels_per_gpu"],
                 "samples_per_channel": w["cfg"]["samples_per_channel_per_step"],
                 "Msamples_per_s": w["units"] * a.steps / dt / 1e6, "launch_ms": launch_s * 1e3, "kernel": kernel,
                 "frac_hbm": w["bytes_per_sample"] * w["units"] / launch_s / 1e9 / HBM_PEAK_GBS,
                 "algorithmic_bytes_per_sample": w["bytes_per_sample"]}
            if w["report"]:
                e.update(w["report"]())
            comp = roof.get("compute") or (roof if roof["bound"] == "mfma" else None)
            if comp:
                e["compute"] = {k: comp[k] for k in ("achieved", "peak", "unit", "frac") if k in comp}
                if "peak_is" in comp:
                    e["compute"]["peak_is"] = comp["peak_is"]
            out[name] = e
            del w
        except Exception as ex:  # an extra must never cost the headline line
            out[name] = {"error": repr(ex)}
        torch.cuda.empty_cache()
    return out


def host_layer(torch, dev, L):
    """The drop-in boundary as its callers meet it (rtlsdr_receiver.cpp:1974-2116, ProfileTest/main.cpp:129-191): HOST
    buffers in, host buffers out, every call synchronous; PCIe-inclusive, so never the headline `value`.  Each entry:
    Msamples/s, ms per call, host bytes moved per second, and the share of the measured pinned-copy rate that is."""
    import numpy as np

    from sdr_amd import device, host
    out = {}
    rng = np.random.default_rng(11)

    def timed(fn, min_s=0.6, max_calls=50):
        fn()  # warm-up: staging buffers, streams
        k, t0 = 0, time.perf_counter()
        while True:
            fn()
            k += 1
            dt = time.perf_counter() - t0
            if dt >= min_s or k >= max_calls:
                return dt / k

    # the PCIe ceiling on this box: pinned host <-> device copies of 256 MiB
    pin = torch.empty(1 << 28, dtype=torch.uint8).pin_memory()
    d = torch.empty(1 << 28, dtype=torch.uint8, device=dev)

    def h2d():
        d.copy_(pin, non_blocking=True)
        torch.cuda.synchronize()

    def d2h():
        pin.copy_(d, non_blocking=True)
        torch.cuda.synchronize()
    h2d_gbs, d2h_gbs = (1 << 28) / timed(h2d, 0.3) / 1e9, (1 << 28) / timed(d2h, 0.3) / 1e9
    del pin, d
    out["pcie"] = {"h2d_pinned_GBps": h2d_gbs, "d2h_pinned_GBps": d2h_gbs,
                   "note": "torch pinned <-> device copies of 256 MiB on this box: the ceiling of every entry below"}

    def entry(name, what, n_samples, bytes_in, bytes_out, sec):
        floor = bytes_in / (h2d_gbs * 1e9) + bytes_out / (d2h_gbs * 1e9)  # both directions at the measured copy rates, back to back
        out[name] = {"what": what, "samples_per_call": n_samples, "ms_per_call": sec * 1e3, "Msamples_per_s": n_samples / sec / 1e6,
                     "host_GBps": (bytes_in + bytes_out) / sec / 1e9, "bytes_in": bytes_in, "bytes_out": bytes_out,
                     "pcie_floor_ms": floor * 1e3, "frac_of_pcie_floor": floor / sec}

    try:
        n = 1 << 25
        taps = lpf_taps(64, 2.4e6, 1.0e5)
        re = rng.integers(-127, 128, n).astype(np.float64)
        im = rng.integers(-127, 128, n).astype(np.float64)
        ch = host.Chain(taps, 10)
        entry("chain_fir_fmdisc_f64", "host.Chain.fir_fmdisc = sdrx_chain_fir_fmdisc: cfg2 chain on two fp64 host rails of 2^25 samples",
              n, 16 * n, 8 * (n // 10), timed(lambda: ch.fir_fmdisc(re, im)))
        raw = rng.integers(0, 256, (n, 2)).astype(np.uint8)
        entry("chain_fir_fmdisc_u8", "sdrx_chain_fir_fmdisc_u8: the same chain from the radio's raw uint8 {I,Q} bytes (converted on the device)",
              n, 2 * n, 8 * (n // 10), timed(lambda: ch.fir_fmdisc_u8(raw)))
        f = host.Filter(host.LPF, 64, 2.4e6, 1.0e5)
        buf = re.copy()
        entry("filter_execute", "host.Filter.execute = Filter::execute (filt.cpp:267-273), in place on 2^25 doubles",
              n, 8 * n, 8 * n, timed(lambda: f.execute(buf)))
        del ch, raw, f, buf
        nf = 64
        filters = [host.Filter(host.LPF, 64, 2.4e6, 1.0e5) for _ in range(nf)]
        bufs = [re.copy() for _ in range(nf)]  # 16 GiB of host doubles
        entry("fir_execute_batch_64", "sdrx_fir_execute_batch: 64 Filter objects x 2^25 doubles each, in place, one call",
              nf * n, 8 * n * nf, 8 * n * nf, timed(lambda: host.fir_execute_batch(filters, bufs), 0.1, 2))
        del filters, bufs
        # the single-radio sizes BASELINE names
        n1 = 1 << 20
        taps1 = lpf_taps(64, 2.0e6, 1.0e5)
        c1 = host.Chain(taps1, 1)
        r1, i1 = re[:n1].copy(), im[:n1].copy()
        entry("cfg1_one_2^20_buffer", "host.Chain.fir_am: BASELINE configs[0] as stated, ONE 2^20-sample I/Q buffer per call (host in, host bits out)",
              n1, 16 * n1, n1, timed(lambda: c1.fir_am(r1, i1), 0.3, 200))
        del c1
    except Exception as ex:
        out["error"] = repr(ex)[:300]
    try:  # cfg2 as the receiver streams it: 2^18-sample chunks, state carried, 8 chunk launches replayed from one HIP graph
        chunks, nchunk = 8, 1 << 18
        taps = lpf_taps(64, 2.4e6, 1.0e5)
        x = synth_fm_stream(torch, chunks * nchunk, 77, dev).view(chunks, nchunk, 2)
        chain = device.FirChain(taps, nchan=1, decim=10, device=dev)
        o = torch.zeros(chunks, chain.n_out(nchunk), dtype=torch.float64, device=dev)
        ring = device.FirRing(chain, x, o, epilogue="fmdisc", use_graph=2)

        def replay():
            ring.launch()
            torch.cuda.synchronize()
        sec = timed(replay, 0.3, 500)
        out["cfg2_ring_of_2^18_chunks"] = {"what": "sdrx_fir_ring_*: cfg2 in 2^18-sample chunks (device-resident ring of 8, state carried, one HIP "
                                                   "graph replay + synchronize per 8 chunks)", "samples_per_call": chunks * nchunk,
                                           "ms_per_call": sec * 1e3, "us_per_chunk": sec / chunks * 1e6,
                                           "Msamples_per_s": chunks * nchunk / sec / 1e6,
                                           "frac_hbm": 16.8 * chunks * nchunk / sec / 1e9 / HBM_PEAK_GBS}
        del ring, chain, x, o
    except Exception as ex:
        out["cfg2_ring_error"] = repr(ex)[:300]
    torch.cuda.empty_cache()
    return out


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        import torch  # device_count() does not initialise the GPU on this image; nothing else here touches it
        have = torch.cuda.device_count()
        if have < args.gpus and not os.environ.get("SDRX_BENCH_SHARE_GPU"):
            raise SystemExit(f"bench.py: --gpus {args.gpus} but {have} visible (SDRX_BENCH_SHARE_GPU=1 --backend gloo "
                             "rehearses the multi-rank path on one GPU)")
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))
    worker(args)


if __name__ == "__main__":
    main()
