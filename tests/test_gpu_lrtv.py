"""GPU parity: the LRTV solver option (FISTA_deep.m, unlocbox prox_tv / norm_tv) through the C ABI vs the CPU oracle.
fp64 on both sides; the only differences are summation orders, so images agree to ~1e-12 and the iteration counts of the
inner (tolerance 10e-4 on the relative objective change) and outer loops are identical."""
import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu


def _image(R, C, seed):
    rng = np.random.default_rng(seed)
    steps = np.add.outer(np.arange(R) // max(R // 5, 1), np.arange(C) // max(C // 7, 1)).astype(np.float64)
    return steps + 0.3 * rng.standard_normal((R, C))


@pytest.mark.parametrize("shape", [(64, 16), (65, 17), (40, 52), (130, 33), (448, 2240)])
def test_norm_tv_and_prox_tv_vs_oracle(engine_mod, oracle, shape):
    R, C = shape
    b = _image(R, C, seed=R * 1000 + C)
    e = engine_mod.Engine(0)
    n_g, n_o = e.norm_tv(b), oracle.norm_tv(b)
    assert abs(n_g - n_o) <= 1e-12 * n_o
    for gamma in (0.05, 0.7, 4.0):
        sg, ig, og = e.prox_tv(b, gamma)
        so, io, oo = oracle.prox_tv(b, gamma)
        assert ig == io, (gamma, ig, io)
        assert np.max(np.abs(sg - so)) <= 1e-11 * np.max(np.abs(so))
        assert abs(og - oo) <= 1e-11 * oo
    e.close()


def test_prox_tv_edge_cases(engine_mod, oracle):
    e = engine_mod.Engine(0)
    b = _image(48, 40, seed=5)
    s0, i0, _ = e.prox_tv(b, 0.0)                                     # test_gamma: gamma = 0 returns the input, 0 iterations
    assert i0 == 0 and np.array_equal(s0, b)
    sg, ig, _ = e.prox_tv(b, 0.5, tol=1e-9, maxit=7)                 # maxit reached: sol of the last iteration
    so, io, _ = oracle.prox_tv(b, 0.5, tol=1e-9, maxit=7)
    assert ig == io == 7 and np.max(np.abs(sg - so)) < 1e-11
    const = np.full((20, 24), 3.25)                                   # constant image: fixed point
    sc, ic, oc = e.prox_tv(const, 1.0)
    assert np.array_equal(sc, const)
    big, _, _ = e.prox_tv(b, 1e4)                                     # huge gamma: towards the mean image
    bo, _, _ = oracle.prox_tv(b, 1e4)
    assert np.max(np.abs(big - bo)) < 1e-9
    with pytest.raises(engine_mod.QmriError):
        e.prox_tv(b, -1.0)                                           # unlocbox test_gamma errors on negative gamma
    with pytest.raises(ValueError):
        e.prox_tv(np.zeros((4, 4, 2)), 1.0)
    e.close()


def test_tv_operator_mirror(oracle):
    from qmri_pnp_recon_poc_amd import reference_api as R
    J = R.TV_operator("2D", 0)
    x = np.stack([_image(32, 24, 1), _image(32, 24, 2)], axis=2)
    try:
        assert abs(J.norm(x) - (oracle.norm_tv(x[:, :, 0]) + oracle.norm_tv(x[:, :, 1]))) < 1e-9
        p = J.prox(x, 0.3)
        for i in range(2):
            assert np.max(np.abs(p[:, :, i] - oracle.prox_tv(x[:, :, i], 0.3)[0])) < 1e-11
    finally:
        R.release()


def _small(oracle, synth, N=32, T=24, s=6, S=120, seed=0):
    dic = synth.make_dictionary(T=T, n_t1=24, n_t2=16, s=s)
    X0 = synth.synthesize_tsmi(synth.make_phantom_qmaps(N, seed=seed), dic)
    fp, k = oracle.spiral_mask(N, S, T)
    op = oracle.Operator(N, N, dic["V"], fp, k)
    y = synth.awgn_measured(op.forward(X0), 30.0, seed=seed)
    return dic, X0, fp, k, op, y


def test_lrtv_small_vs_oracle(engine_mod, oracle, synth):
    dic, X0, fp, k, op, y = _small(oracle, synth)
    e = engine_mod.Engine(0)
    with pytest.raises((engine_mod.QmriError, ValueError)):
        e.lrtv(y)                                                     # operator not set
    e.set_operator(32, 32, dic["V"], fp, k)
    for K in (4e-5, 2e-3, 0.0):                                       # the script's K, a K that makes the prox work, no prox at all
        xg, ig = e.lrtv(y, K=K, iters=25)
        xo, io = oracle.fista_lrtv(op, y, K=K, iters=25)
        print(f"K={K}: iters {ig['iters']}/{io['iters']} halvings {ig['halvings']}/{io['halvings']} prox iters {ig['prox_iters_total']}/{int(io['prox_iters'].sum())} "
              f"rel_err {rel_err(xg, xo):.2e}")
        assert ig["iters"] == io["iters"] and ig["halvings"] == io["halvings"]
        assert ig["prox_iters_total"] == int(io["prox_iters"].sum())
        assert abs(ig["step"] - io["step"]) == 0.0
        assert rel_err(xg, xo) < 1e-9
        assert abs(ig["obj"] - io["obj"][-1]) <= 1e-10 * abs(io["obj"][-1])
    xg, ig = e.lrtv(y, K=2e-3, iters=6, backtrack=False)
    xo, io = oracle.fista_lrtv(op, y, K=2e-3, iters=6, backtrack=False)
    assert ig["halvings"] == 0 and rel_err(xg, xo) < 1e-9
    e.close()


def test_lrtv_224_vs_oracle(engine_mod, oracle, synth, case224):
    e = engine_mod.Engine(0)
    e.set_operator(224, 224, case224["dic"]["V"], case224["fp"], case224["k"])
    xg, ig = e.lrtv(case224["y"], iters=8)                            # the script's parameters (:274-279), first 8 iterations
    xo, io = oracle.fista_lrtv(case224["op"], case224["y"], iters=8)
    print(f"224: iters {ig['iters']} halvings {ig['halvings']} prox iters {ig['prox_iters_total']} rel_err {rel_err(xg, xo):.2e}")
    assert ig["iters"] == io["iters"] and ig["halvings"] == io["halvings"] and ig["prox_iters_total"] == int(io["prox_iters"].sum())
    assert rel_err(xg, xo) < 1e-9
    e.close()
