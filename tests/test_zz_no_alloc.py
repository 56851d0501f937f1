"""sdrx_psk_demod_noalloc_dev -- the one entry point that allocates nothing (include/sdrx.h) -- against the entry
point it is a variant of, sdrx_psk_demod_dev, through the C ABI.  Bar: bit-identical decisions and loop state
(integer decisions; the state doubles come out of the same kernels).  Two calls in a row on one set of states,
the second with a smaller input (fewer channels AND fewer samples), so that whatever the first call left in the
static scratch array would show; then the limit and the no-allocation claim itself.

Collected last on purpose (the round's end runs pytest -x)."""
import ctypes as C

import numpy as np
import pytest

import _oracle as O
from signals import psk_test_signal

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from sdr_amd import _lib, device  # noqa: E402


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    device.init(0)
    yield
    torch.cuda.synchronize()


def _bank_input(kind, nchan, n, seed):
    iqs = [psk_test_signal(kind, n // 8, seed=seed + c, phase_deg=37.0 * c, snr_db=18.0, timing_offset=c % 5)
           for c in range(nchan)]
    return torch.tensor(np.stack([np.asarray(z).reshape(-1, 2) for z in iqs]), device="cuda")


def _ptr(t):
    return C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


@pytest.mark.parametrize("kind,lk", [(O.QPSK, _lib.QPSK), (O.BPSK, _lib.BPSK)])
def test_noalloc_equals_workspace_entry_point_two_calls(kind, lk):
    L = _lib.lib()
    shapes = [(70, 4096), (9, 1000)]  # second call: smaller in both directions, ragged length
    ref = device.PskBank(lk, shapes[0][0], shapes[0][1])       # sdrx_psk_demod_dev, caller workspace
    st_ref = ref.states
    st_new = ref.states.clone()                                # the same initial states for the new entry point
    for call, (nchan, n) in enumerate(shapes):
        x = _bank_input(kind, nchan, n, seed=500 + 100 * call)
        n_dec = (n // 4 + 1) // 2
        want = torch.full((nchan, n_dec + 3), 0xEE, dtype=torch.uint8, device="cuda")
        got = torch.full((nchan, n_dec + 3), 0xEE, dtype=torch.uint8, device="cuda")
        _lib.check(L.sdrx_psk_demod_dev(lk, _ptr(x), n, nchan, n, _ptr(st_ref), _ptr(want), want.shape[1], _ptr(ref.work),
                                        _stream()), "sdrx_psk_demod_dev")
        torch.cuda.synchronize()
        free0 = torch.cuda.mem_get_info()[0]
        _lib.check(L.sdrx_psk_demod_noalloc_dev(lk, _ptr(x), n, nchan, n, _ptr(st_new), _ptr(got), got.shape[1], _stream()),
                   "sdrx_psk_demod_noalloc_dev")
        torch.cuda.synchronize()
        assert torch.cuda.mem_get_info()[0] == free0, "device memory was allocated or freed by the call"
        assert torch.equal(got, want), (call, nchan, n)
        assert torch.equal(got[:, n_dec:], torch.full_like(got[:, n_dec:], 0xEE))  # nothing written past n/8 decisions
        assert torch.equal(st_new, st_ref), (call, "loop state / confidence counters")
        # and both equal the CPU oracle (first channel is enough here: the bank tests cover the rest)
        if call == 0:
            st = O.psk_new(kind, math_mode=O.MATH_SHARED)
            o = O.psk_demodulate(st, x[0].cpu().numpy().reshape(-1))
            assert np.array_equal(got[0, :n_dec].cpu().numpy(), o.astype(np.uint8))


def test_noalloc_rejects_sizes_beyond_the_static_array():
    L = _lib.lib()
    limit = L.sdrx_psk_demod_noalloc_max_bytes()
    assert 0 < limit < 16 << 20
    nchan, n = 64, 1 << 16  # 64 x 2^14 complex doubles = 16 MiB of matched-filter output alone
    assert L.sdrx_psk_work_bytes(n, nchan) > limit
    x = torch.zeros(1, 8, 2, dtype=torch.float64, device="cuda")  # never read: the size check comes first
    st = device.PskBank(_lib.QPSK, 1, 8).states
    out = torch.zeros(1, 8, dtype=torch.uint8, device="cuda")
    rc = L.sdrx_psk_demod_noalloc_dev(_lib.QPSK, _ptr(x), n, nchan, n, _ptr(st), _ptr(out), n // 8, _stream())
    assert rc == -3  # SDRX_EUNSUPPORTED
    assert b"static scratch" in L.sdrx_last_error()
    # the largest shape that fits still runs
    nchan, n = 16, 1 << 15
    assert L.sdrx_psk_work_bytes(n, nchan) <= limit
    x = torch.randn(nchan, n, 2, dtype=torch.float64, device="cuda")
    bank = device.PskBank(_lib.QPSK, nchan, n)
    st2 = bank.states.clone()
    want = torch.zeros(nchan, n // 8, dtype=torch.uint8, device="cuda")
    got = torch.zeros_like(want)
    bank.demod(x, want)
    _lib.check(L.sdrx_psk_demod_noalloc_dev(_lib.QPSK, _ptr(x), n, nchan, n, _ptr(st2), _ptr(got), n // 8, _stream()),
               "sdrx_psk_demod_noalloc_dev")
    torch.cuda.synchronize()
    assert torch.equal(got, want) and torch.equal(st2, bank.states)


def test_noalloc_runs_under_stream_capture():
    """The callers it is for: a HIP graph capture forbids hipMalloc and synchronisation.  Capture one call, replay
    it twice, compare with the workspace entry point."""
    L = _lib.lib()
    nchan, n = 33, 2048
    x = _bank_input(O.QPSK, nchan, n, seed=900)
    bank = device.PskBank(_lib.QPSK, nchan, n)
    st_cap = bank.states.clone()
    n_dec = bank.n_dec
    got = torch.zeros(nchan, n_dec, dtype=torch.uint8, device="cuda")
    want = torch.zeros_like(got)
    s = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        _lib.check(L.sdrx_psk_demod_noalloc_dev(_lib.QPSK, _ptr(x), n, nchan, n, _ptr(st_cap), _ptr(got), n_dec,
                                                C.c_void_p(torch.cuda.current_stream().cuda_stream)),
                   "sdrx_psk_demod_noalloc_dev under capture")
    for _ in range(2):  # state carries from replay to replay exactly as from call to call
        g.replay()
        bank.demod(x, want)
        torch.cuda.synchronize()
        assert torch.equal(got, want) and torch.equal(st_cap, bank.states)
