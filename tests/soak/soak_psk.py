"""One-off soak (not part of the suite): random PSK banks on the GPU against the oracle, decisions + loop state +
counters, over kinds, unique-word lengths, SNRs down to noise, amplitude scales, frequency offsets, ragged lengths."""
import sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
import _oracle as O
from signals import psk_test_signal
from sdr_amd import _lib, device
device.init(0)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
t_end = time.time() + float(sys.argv[2] if len(sys.argv) > 2 else 60)
cases = bad = 0
t_say = time.time() + 30


def same_bits(a, b):  # the same bits (signs of zeros included), or NaN in the same places
    a, b = np.ascontiguousarray(a, dtype=np.float64), np.ascontiguousarray(b, dtype=np.float64)
    return bool(np.all((a.view(np.uint64) == b.view(np.uint64)) | (np.isnan(a) & np.isnan(b))))


while time.time() < t_end:
    if time.time() > t_say:  # (a run that prints nothing for minutes is taken to be hung)
        print("soak: %d channel-blocks so far, %d mismatches" % (cases, bad), flush=True)
        t_say = time.time() + 30
    kind, lk = ((O.BPSK, _lib.BPSK), (O.QPSK, _lib.QPSK))[int(rng.integers(0, 2))]
    n = int(rng.choice([512, 1000, 1024, 2048, 4096, 4096 + 8 * 3, 4100, 8192, 16384]))
    nchan = int(rng.choice([1, 3, 64, 65, 130]))
    blocks = int(rng.integers(1, 4))
    if kind == O.BPSK:
        U = int(rng.choice([8, 13, 16, 64])); uw = int(rng.integers(0, 2 ** 63)) & ((1 << U) - 1) if U < 64 else int(rng.integers(0, 2 ** 63))
        mk = lambda: O.psk_new(kind, uw=uw, uw_len=U, math_mode=O.MATH_SHARED)
        bank = device.PskBank(lk, nchan, n, unique_word=uw, uw_len=U)
    else:
        U = int(rng.choice([4, 13, 16, 64])); uwq = np.array(rng.integers(0, 4, 64), dtype=np.float64)
        mk = lambda: O.psk_new(kind, uw=uwq, uw_len=U, math_mode=O.MATH_SHARED)
        bank = device.PskBank(lk, nchan, n, unique_word=uwq, uw_len=U)
    sts = [mk() for _ in range(nchan)]
    for blk in range(blocks):
        iqs = []
        for c in range(nchan):
            snr = float(rng.choice([-10.0, 0.0, 6.0, 12.0, 25.0, 60.0]))
            z = np.asarray(psk_test_signal(kind, (n + 7) // 8, seed=int(rng.integers(0, 1 << 30)), phase_deg=float(rng.uniform(0, 360)),
                                           snr_db=snr, timing_offset=int(rng.integers(0, 8)), freq_offset=float(rng.choice([0, 1e-4, -3e-3, 2e-2])),
                                           uw_bpsk=(uw if kind == O.BPSK else 0x4E), uw_len=(U if kind == O.BPSK else None)))[: 2 * n]
            z = z * float(rng.choice([1.0, 1e-3, 50.0, 1e6]))
            z = np.array(z, dtype=np.float64)
            r = rng.random()
            if r < 0.12:  # a silent stretch (exact zeros reach the loops: the speculative blocks' replay)
                a0 = int(rng.integers(0, n)); z[2 * a0: 2 * min(n, a0 + int(rng.integers(1, 1500)))] = 0.0
            elif r < 0.15:
                z[:] = 0.0 if rng.random() < 0.5 else -0.0
            elif r < 0.18:
                z[int(rng.integers(0, 2 * n))] = float(rng.choice([np.nan, np.inf, -np.inf]))
            elif r < 0.21:
                z[int(rng.integers(0, 2))::2] = 0.0  # one rail silent
            iqs.append(z)
        x = torch.tensor(np.stack([z.reshape(-1, 2) for z in iqs]), device="cuda")
        out = torch.zeros(nchan, bank.n_dec + 2, dtype=torch.uint8, device="cuda")
        bank.demod(x, out)
        got = out.cpu().numpy(); hs = bank.host_states()
        for c in range(nchan):
            want = O.psk_demodulate(sts[c], iqs[c])
            ok = np.array_equal(got[c][: want.size], want) and list(hs[c].cc) == list(sts[c].cc) and \
                same_bits(np.frombuffer(bytes(hs[c])[: 8 * 24], dtype=np.float64), sts[c].loop_state())
            cases += 1
            if not ok:
                bad += 1
                print("MISMATCH", kind, n, nchan, U, blk, c, flush=True)
print("soak: %d channel-blocks, %d mismatches" % (cases, bad), flush=True)
sys.exit(1 if bad else 0)
