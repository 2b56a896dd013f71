"""Decision injection, checked on the oracle alone (CPU): ReLU masks and max-pool taps recorded from the float32 oracle
and injected into the float64 oracle make the two differentiate the same piecewise-linear branch - the float32
gradients then agree with float64 to smooth rounding, an order of magnitude closer than against the free float64 run
(where every flipped unit moves all upstream gradients).  This is the mechanism tests/test_fullsize_gpu.py uses with the
HIP forward's decisions; here it is pinned without a GPU, through the same worker processes."""
import numpy as np

from helpers import run_oracle_jobs
from oracle_worker import pack_decisions, unpack_decisions


def test_pack_roundtrip(tmp_path):
    rng = np.random.default_rng(0)
    store = {"a.bn1": [rng.random((2, 5, 7, 3)) > 0.5, rng.random((1, 3)) > 0.5],
             "enc.maxpool": [rng.integers(0, 9, (2, 4, 3, 3)).astype(np.int8)]}
    p = str(tmp_path / "d.npz")
    pack_decisions(store, p)
    back = unpack_decisions(p)
    assert set(back) == set(store)
    for k in store:
        assert len(back[k]) == len(store[k])
        for a, b in zip(store[k], back[k]):
            assert a.shape == b.shape and np.array_equal(a, b)


def test_injected_decisions_remove_the_branch_noise(tmp_path):
    base = dict(kind="clip_ocr", arch="resnet50", T=3, B=2, S=65, full_grads=True)
    dec = str(tmp_path / "dec.npz")
    r32, free, inj = run_oracle_jobs(
        [dict(base, dtype="f32", decisions="record", decisions_path=dec, out=str(tmp_path / "r32.npz")),
         dict(base, dtype="f64", out=str(tmp_path / "free64.npz")),
         dict(base, dtype="f64", decisions="inject", decisions_path=dec, out=str(tmp_path / "inj64.npz"), after=0)],
        str(tmp_path), parallel=2, threads=4)
    names = [str(n) for n in r32["names"]]
    scale = float(inj["norms"].max())

    def rel(ref):
        out = []
        for n in names:
            d = np.linalg.norm(r32["g:" + n].astype(np.float64) - ref["g:" + n].astype(np.float64))
            out.append(d / max(float(np.linalg.norm(ref["g:" + n].astype(np.float64))), 1e-3 * scale))
        return np.array(out)

    e_free, e_inj = rel(free), rel(inj)
    print("float32 vs float64 free: median %.2e max %.2e | injected: median %.2e max %.2e"
          % (np.median(e_free), e_free.max(), np.median(e_inj), e_inj.max()))
    assert abs(float(r32["loss"]) - float(inj["loss"])) < 1e-6 * float(inj["loss"])
    assert np.median(e_inj) < 3e-3 and e_inj.max() < 6e-3          # measured 1.3e-3 / 1.5e-3
    assert np.median(e_free) > 5 * np.median(e_inj)                 # measured 3.1e-2 vs 1.3e-3


def test_sequential_gemm_is_one_fmaf_chain_per_element():
    """oracle/csrc/seq_gemm.c against an explicit k-ordered chain (products of two float32 are exact in float64, the
    single rounding of each step is the cast back), ragged sizes and the chunked (split-K) form."""
    from oracle import np_ops as O

    rng = np.random.default_rng(3)
    a = rng.standard_normal((13, 301)).astype(np.float32)
    b = rng.standard_normal((21, 301)).astype(np.float32)

    def chain(k0, k1):
        acc = np.zeros((13, 21), dtype=np.float32)
        for k in range(k0, k1):
            acc = (a[:, k:k + 1].astype(np.float64) * b[None, :, k].astype(np.float64) + acc.astype(np.float64)) \
                .astype(np.float32)
        return acc

    O.set_gemm("sequential")
    try:
        got = O._mm_nt(a, b)
        got_c = O._mm_nt(a, b, chunk=128)
        batched = O._mm_nt(np.stack([a, a[::-1]]), np.stack([b, b]))
    finally:
        O.set_gemm("blas")
    assert np.array_equal(got, chain(0, 301))
    want = chain(0, 128)
    want = want + chain(128, 256)
    want = want + chain(256, 301)
    assert np.array_equal(got_c, want)
    assert np.array_equal(batched[0], got) and np.array_equal(batched[1], got[::-1])
