"""GPU parity: the PnP-ADMM loop through the C ABI vs the CPU oracle (PnP_ADMM.m:1-148)."""
import ctypes as C

import numpy as np
import pytest

from conftest import assert_atoms_close, atom_tolerance, rel_err

pytestmark = pytest.mark.gpu


def tsmi_psnr(a, b):
    """mean per-channel PSNR on abs with peak 1 (main_recon_tsmis_FFT.m:362-367)."""
    return float(np.mean([10 * np.log10(1.0 / max(np.mean((np.abs(a[:, :, c]) - np.abs(b[:, :, c])) ** 2), 1e-300)) for c in range(a.shape[2])]))


def test_admm_224_single_level_vs_oracle(engine_mod, oracle, synth, case224):
    w = synth.structured_weights(seed=2, eps=0.02)
    e = engine_mod.Engine(0)
    e.set_operator(224, 224, case224["dic"]["V"], case224["fp"], case224["k"])
    e.set_denoiser(w, 224, 224)
    iters = 8
    xg, dg, lg = e.pnp_admm(case224["y"], iters=iters, gt=case224["X0"], want_diag=True)
    net = oracle.Net(w)
    xo, do, lo = oracle.pnp_admm(case224["op"], net, case224["y"], iters=iters, gt=case224["X0"], want_diag=True)
    # Tolerances: data consistency is fp64 on both sides, the network fp32 with different summation orders
    # (MFMA k-chains vs CPU SIMD), so x agrees to ~1e-5 relative; the stated bound is 1e-4 relative L2 and
    # >= 80 dB mean TSMI PSNR between the two reconstructions.
    same_counts = bool(np.array_equal(lg, lo))
    err, psnr = rel_err(xg, xo), tsmi_psnr(xg, xo)
    diag_ok = bool(np.allclose(dg, do, rtol=1e-4, atol=0))
    print(f"224 single_level: lsqr {lg.tolist()} rel_err {err:.3e} psnr_vs_oracle {psnr:.1f} dB")
    assert same_counts                                              # identical LSQR iteration counts
    assert err < 1e-4
    assert psnr > 80.0
    assert diag_ok
    # (PSNR against the ground truth is not asserted: the synthetic weights are not a trained denoiser.)
    # DIRECT solver: exact minimiser; lands within the LSQR stop-rule ambiguity of the reference path
    xd, _, _ = e.pnp_admm(case224["y"], iters=iters, solver="direct")
    xod, _, _ = oracle.pnp_admm(case224["op"], net, case224["y"], iters=iters, solver="direct")
    e1, e2 = rel_err(xd, xod), rel_err(xd, xg)
    print(f"direct vs oracle-direct {e1:.3e}; direct vs lsqr {e2:.3e}")
    assert e1 < 1e-4
    assert e2 < 2e-3
    e.close()


def test_admm_224_with_an_interior_that_matters(engine_mod, oracle, synth, case224):
    """The 224 x 224 loop with a network whose interior layers carry weight (round 3): structured_weights(eps=0.02), the stable
    bench network, is numerically head + tail, so the ADMM parity runs above cannot see the deep-level kernels.  With eps = 0.5
    the loop is still bounded over a few iterations (max|x| ~ 4) and one interior layer times 1.01 moves x by 4e-4 ... 2e-3
    (asserted on the oracle first: the guard that keeps this test from going blind); GPU vs oracle must agree to 1e-4."""
    iters = 4
    w = synth.structured_weights(seed=2, eps=0.5)
    xo, _, lo = oracle.pnp_admm(case224["op"], oracle.Net(w), case224["y"], iters=iters)
    for name in ("m_body.1.res.2.weight", "m_down1.2.res.0.weight"):
        w2 = w.copy()
        w2[synth.unetres_weight_slice(name)] *= np.float32(1.01)
        x2, _, _ = oracle.pnp_admm(case224["op"], oracle.Net(w2), case224["y"], iters=iters)
        moved = rel_err(x2, xo)
        print(f"sensitivity guard: {name} x 1.01 moves x after {iters} iterations by {moved:.2e}")
        assert moved > 2e-4, f"the ADMM parity network went blind to {name}"
    e = engine_mod.Engine(0)
    e.set_operator(224, 224, case224["dic"]["V"], case224["fp"], case224["k"])
    e.set_denoiser(w, 224, 224)
    xg, _, lg = e.pnp_admm(case224["y"], iters=iters)
    err = rel_err(xg, xo)
    print(f"224 eps=0.5: lsqr gpu {lg.tolist()} oracle {lo.tolist()}, rel_err {err:.3e}, max|x| {np.abs(xo).max():.3g}, scheme {e.denoiser_scheme()}")
    assert np.array_equal(lg, lo)
    assert err < 1e-4
    e.close()


def _small_case(oracle, synth, N=32, T=24, s=6, S=120, in_extra=0, seed=0):
    dic = synth.make_dictionary(T=T, n_t1=24, n_t2=16, s=s)
    q = synth.make_phantom_qmaps(N, seed=seed)
    X0 = synth.synthesize_tsmi(q, dic)
    fp, k = oracle.spiral_mask(N, S, T)
    op = oracle.Operator(N, N, dic["V"], fp, k)
    y = synth.awgn_measured(op.forward(X0), 30.0, seed=seed)
    nc = (8, 16, 16, 32)
    w = synth.structured_weights(in_nc=s + in_extra, out_nc=s, nc=nc, nb=2, seed=3, eps=0.05)
    return dic, X0, fp, k, op, y, nc, w


@pytest.mark.parametrize("multi", [False, True])
def test_admm_small_100_iterations(engine_mod, oracle, synth, multi):
    dic, X0, fp, k, op, y, nc, w = _small_case(oracle, synth, in_extra=1 if multi else 0)
    s = X0.shape[2]
    e = engine_mod.Engine(0)
    e.set_operator(32, 32, dic["V"], fp, k)
    e.set_denoiser(w, 32, 32, in_nc=s + (1 if multi else 0), out_nc=s, nc=nc, nb=2)
    xg, dg, lg = e.pnp_admm(y, iters=100, multi_level=multi, noise_std=0.01, gt=X0, want_diag=True)
    net = oracle.Net(w, in_nc=s + (1 if multi else 0), out_nc=s, nc=nc, nb=2)
    xo, do, lo = oracle.pnp_admm(op, net, y, iters=100, multi_level=multi, noise_std=0.01, gt=X0, want_diag=True)
    frac, err = float(np.mean(lg == lo)), rel_err(xg, xo)
    maxdiff = int(np.abs(lg - lo).max())
    diag_ok = bool(np.allclose(dg[:, 0], do[:, 0], rtol=5e-3))
    print(f"small multi={multi}: same-count fraction {frac:.2f}, max count diff {maxdiff}, rel_err {err:.3e}")
    # The LSQR stop test is a threshold on a continuous quantity: late in the run (1-3 inner iterations) the
    # fp32 network's summation-order differences flip it by one now and then (measured: 8-24 % of the 100 x-updates,
    # once by two).  x itself stays far inside the stop-rule ambiguity (~2.2e-4 relative per
    # x-update; measured 2e-5 .. 5e-5 after 100 iterations).
    assert frac > 0.7 and maxdiff <= 2
    assert err < 2e-4
    assert diag_ok
    e.close()


def test_admm_x0_iters0_and_errors(engine_mod, oracle, synth):
    dic, X0, fp, k, op, y, nc, w = _small_case(oracle, synth)
    s = X0.shape[2]
    e = engine_mod.Engine(0)
    with pytest.raises((engine_mod.QmriError, ValueError)):
        e.pnp_admm(np.zeros(3, complex))                            # operator not set
    e.set_operator(32, 32, dic["V"], fp, k)
    with pytest.raises(engine_mod.QmriError):
        e.pnp_admm(y, iters=1)                                       # denoiser not set
    e.set_denoiser(w, 32, 32, in_nc=s, out_nc=s, nc=nc, nb=2)
    x, _, _ = e.pnp_admm(y, iters=0)                                 # zero iterations returns X0 = F.adjoint(y)
    assert rel_err(x, op.adjoint(y)) < 1e-12
    xi = 0.5 * op.adjoint(y)
    xg, _, _ = e.pnp_admm(y, iters=3, x0=xi)
    xo, _, _ = oracle.pnp_admm(op, oracle.Net(w, in_nc=s, out_nc=s, nc=nc, nb=2), y, iters=3, x0=xi)
    assert rel_err(xg, xo) < 1e-4
    with pytest.raises(engine_mod.QmriError):
        e.pnp_admm(y, iters=1, multi_level=True)                     # 6-channel net cannot take the noise map
    e.close()


def test_recon_batch_single_device(engine_mod, oracle, synth):
    from qmri_pnp_recon_poc_amd import batch
    dic, X0, fp, k, op, y0, nc, w = _small_case(oracle, synth)
    s = X0.shape[2]
    ys = []
    for sl in range(3):
        q = synth.make_phantom_qmaps(32, seed=sl)
        ys.append(synth.awgn_measured(op.forward(synth.synthesize_tsmi(q, dic)), 30.0, seed=sl))
    ys = np.stack(ys)
    res = batch.recon_batch([0], ys, N=32, M=32, V=dic["V"], frame_ptr=fp, kidx=k, weights=w, in_nc=s, out_nc=s, nc=nc, nb=2,
                            dictionary=dic, iters=5, slices_per_launch=2)
    net = oracle.Net(w, in_nc=s, out_nc=s, nc=nc, nb=2)
    for sl in range(3):
        xo, _, _ = oracle.pnp_admm(op, net, ys[sl], iters=5)
        err = rel_err(res["X"][sl], xo)
        assert err < 1e-4
        o = oracle.dict_match(res["X"][sl], dic["D"], dic["normD"], dic["lut"])
        same = bool(np.array_equal(res["qmap"][sl], o["qmap"]))       # same X in -> bit-exact maps out
        assert same


def test_config2_epi_multi_level_batch_224(engine_mod, oracle, synth, case224):
    """BASELINE.json configs[2] at full size: cut3 slices, EPI mask (setup_subsampling_epi, 1/65 of the k-space lines per frame),
    the 11-channel multi-level DRUNet with its constant noise-map channel, several slices advanced together."""
    from qmri_pnp_recon_poc_amd import batch
    dic = case224["dic"]
    fp, k = oracle.epi_mask(224, 224, 1 / 65, 200)
    op = oracle.Operator(224, 224, dic["V"], fp, k)
    w = synth.structured_weights(in_nc=11, out_nc=10, seed=5, eps=0.02)
    ys = []
    for sl in range(3):
        X0 = synth.synthesize_tsmi(synth.make_phantom_qmaps(224, seed=10 + sl), dic)
        ys.append(synth.awgn_measured(op.forward(X0), 30.0, seed=10 + sl))
    ys = np.stack(ys)
    iters = 4
    res = batch.recon_batch([0], ys, N=224, M=224, V=dic["V"], frame_ptr=fp, kidx=k, weights=w, in_nc=11, out_nc=10,
                            dictionary=dic, iters=iters, multi_level=True, noise_std=0.01, slices_per_launch=3)
    net = oracle.Net(w, in_nc=11, out_nc=10)
    for sl in range(3):
        xo, _, lo = oracle.pnp_admm(op, net, ys[sl], iters=iters, multi_level=True, noise_std=0.01)
        err = rel_err(res["X"][sl], xo)
        print(f"config2 slice {sl}: rel_err {err:.2e}, oracle lsqr iters {lo.tolist()}")
        assert err < 1e-4


def test_config2_epi_multi_level_batch_of_15_every_slice(engine_mod, oracle, synth, case224):
    """BASELINE.json configs[2] at its batch size: 15 cut3 slices advanced together (the persistent batched kernels on every level),
    EPI mask, 11-channel multi-level DRUNet; EVERY slice against the oracle after two ADMM iterations."""
    dic = case224["dic"]
    fp, k = oracle.epi_mask(224, 224, 1 / 65, 200)
    op = oracle.Operator(224, 224, dic["V"], fp, k)
    w = synth.structured_weights(in_nc=11, out_nc=10, seed=5, eps=0.3)      # eps 0.3: interior layers at the 1e-4 level of x
    ys = np.stack([synth.awgn_measured(op.forward(synth.synthesize_tsmi(synth.make_phantom_qmaps(224, seed=40 + sl), dic)), 30.0, seed=40 + sl)
                   for sl in range(15)])
    from qmri_pnp_recon_poc_amd import batch
    res = batch.recon_batch([0], ys, N=224, M=224, V=dic["V"], frame_ptr=fp, kidx=k, weights=w, in_nc=11, out_nc=10,
                            dictionary=dic, iters=2, multi_level=True, noise_std=0.01, slices_per_launch=15)
    net = oracle.Net(w, in_nc=11, out_nc=10)
    worst = 0.0
    for sl in range(15):
        xo, _, _ = oracle.pnp_admm(op, net, ys[sl], iters=2, multi_level=True, noise_std=0.01)
        err = rel_err(res["X"][sl], xo)
        worst = max(worst, err)
        assert err < 1e-4, (sl, err)
    print(f"config2, 15 slices together: worst rel_err vs oracle {worst:.2e}")


@pytest.mark.parametrize("mask", ["epi", "spiral"])
def test_slice_batches_solve_with_the_one_launch_lsqr_a_slice_or_two_at_a_time(engine_mod, oracle, synth, case224, mask, capfd):
    """Round 5: the x-update of a slice BATCH goes through the one-launch LSQR kernel too -- as many slices per launch as are resident together
    (EPI: one slice's <= 256 one-per-CU units; the spiral: two slices), launch after launch (ks_launch_persist) -- instead of two launches per
    iteration over all slices.  Same arithmetic on the same work units: x of every slice and every LSQR count IDENTICAL to a run of the
    same plan with the two-launch iteration; each slice equal to the oracle's count and to 1e-4 in x.  5 slices: an odd number, so the spiral's last launch holds one slice."""
    dic = case224["dic"]
    fp, k = (oracle.epi_mask(224, 224, 1 / 65, 200) if mask == "epi" else (case224["fp"], case224["k"]))
    op = oracle.Operator(224, 224, dic["V"], fp, k)
    w = synth.structured_weights(seed=2, eps=0.3)
    nsl, iters = 5, 3
    ys = np.stack([synth.awgn_measured(op.forward(synth.synthesize_tsmi(synth.make_phantom_qmaps(224, seed=70 + sl), dic)), 30.0, seed=70 + sl)
                   for sl in range(nsl)])
    out = {}
    for on in (1, 0):
        e = engine_mod.Engine(0)
        e.set_operator(224, 224, dic["V"], fp, k, max_batch=nsl)
        e.set_denoiser(w, 224, 224, max_batch=nsl)
        e.lsqr_persist(bool(on))                                         # (after the plan: both runs work on the same units)
        out[on] = e.pnp_admm_batch(ys, slices_per_launch=nsl, iters=iters)
        e.close()
    assert np.array_equal(out[1][1], out[0][1]), (out[1][1], out[0][1])
    assert np.array_equal(out[1][0], out[0][0]), rel_err(out[1][0], out[0][0])
    # the recovery path with a batch: the test hook makes every slice's first solve time out; the reconstruction repeats itself on the
    # two-launch iteration (one message), same x and counts
    capfd.readouterr()
    e = engine_mod.Engine(0)
    e.set_operator(224, 224, dic["V"], fp, k, max_batch=nsl)
    e.set_denoiser(w, 224, 224, max_batch=nsl)
    e.lsqr_persist(2)
    xt, lt = e.pnp_admm_batch(ys, slices_per_launch=nsl, iters=iters)
    e.close()
    assert capfd.readouterr().err.count("timed out") == 1
    assert np.array_equal(lt, out[0][1]) and np.array_equal(xt, out[0][0])
    net = oracle.Net(w)
    for sl in (0, nsl - 1):
        xo, _, lo = oracle.pnp_admm(op, net, ys[sl], iters=iters)
        assert np.array_equal(out[1][1][sl], np.asarray(lo).ravel()[:iters]), (sl, out[1][1][sl], lo)
        assert rel_err(out[1][0][sl], xo) < 1e-4


def test_config3_thirty_slices_two_workers_224(engine_mod, oracle, synth, case224):
    """BASELINE.json configs[3] at size on one device: 30 cut3 slices at 224 x 224 through qmri_recon_batch with TWO workers (host thread +
    context each; one per GPU on a node, both on device 0 here), 8 slices per launch, 2 ADMM iterations + dictionary match.  Every
    output slot must be filled by its own slice (slot s != slot s'), three sampled slices are compared with the oracle."""
    from qmri_pnp_recon_poc_amd import batch
    dic, op = case224["dic"], case224["op"]
    w = synth.structured_weights(seed=2, eps=0.3)
    nsl = 30
    ys = np.stack([synth.awgn_measured(op.forward(synth.synthesize_tsmi(synth.make_phantom_qmaps(224, seed=100 + sl), dic)), 30.0, seed=100 + sl)
                   for sl in range(nsl)])
    res = batch.recon_batch([0, 0], ys, N=224, M=224, V=dic["V"], frame_ptr=case224["fp"], kidx=case224["k"], weights=w,
                            dictionary=dic, iters=2, slices_per_launch=8)
    X = res["X"]
    assert X.shape == (nsl, 224, 224, 10) and np.isfinite(X).all() and np.isfinite(res["qmap"]).all()
    norms = np.array([np.linalg.norm(X[i]) for i in range(nsl)])
    assert (norms > 0).all() and len(np.unique(np.round(norms, 9))) == nsl        # all slots filled, no slot written twice with one slice
    net = oracle.Net(w)
    for sl in (0, 14, 29):                                                       # first worker's first, the seam, second worker's last
        xo, _, _ = oracle.pnp_admm(op, net, ys[sl], iters=2)
        err = rel_err(X[sl], xo)
        print(f"config3 slice {sl}: rel_err {err:.2e}")
        assert err < 1e-4
        o = oracle.dict_match(X[sl], dic["D"], dic["normD"], dic["lut"])
        assert np.array_equal(res["qmap"][sl], o["qmap"])


def test_cut0_T1000_admm_and_match_at_bench_K(engine_mod, oracle, synth):
    """main_recon_tsmis_FFT.m:41-44 with cut = 0: T = 1000 frames (m = 617 780 spiral samples), end to end: three PnP-ADMM iterations and
    the dictionary match at the bench's K = 98 304, against the oracle; maps bit-exact for the same X."""
    T = 1000
    dic = synth.make_dictionary(T=T, n_t1=384, n_t2=256)
    X0 = synth.synthesize_tsmi(synth.make_phantom_qmaps(224, seed=0), dic)
    fp, k = oracle.spiral_mask(224, 771, T)
    op = oracle.Operator(224, 224, dic["V"], fp, k)
    y = synth.awgn_measured(op.forward(X0), 30.0, seed=0)
    w = synth.structured_weights(seed=2, eps=0.3)
    e = engine_mod.Engine(0)
    e.set_operator(224, 224, dic["V"], fp, k)
    e.set_denoiser(w, 224, 224)
    e.set_dictionary(dic["D"], dic["normD"], dic["lut"])
    xg, _, lg = e.pnp_admm(y, iters=3)
    xo, _, lo = oracle.pnp_admm(op, oracle.Net(w), y, iters=3)
    err = rel_err(xg, xo)
    print(f"cut0 T=1000: m {int(fp[-1])}, lsqr gpu {lg.tolist()} oracle {lo.tolist()}, rel_err {err:.2e}")
    assert np.array_equal(lg, lo) and err < 1e-4
    mg = e.dict_match(xg)
    mx = oracle.dict_match(xg, dic["D"], dic["normD"], dic["lut"])
    assert np.array_equal(mg["dm"], mx["dm"]) and np.array_equal(mg["qmap"], mx["qmap"]) and np.array_equal(mg["pd"], mx["pd"])
    mo = oracle.dict_match(xo, dic["D"], dic["normD"], dic["lut"])
    assert_atoms_close(mg, mo, dic, "cut0")                           # fraction AND distance of the differing pixels on the (T1, T2) grid
    e.close()


def test_config4_cut0_multi_coil_reconstruction_and_match_as_one_pipeline(engine_mod, oracle, synth):
    """BASELINE.json configs[4] as far as it can be defined without a reference counterpart (the reference is single-coil, README.md:63 -- the multi-coil
    parts are a labelled extension, parity unpinned): cut0 (T = 1000, m = 617 780 samples PER COIL), a complex-valued multi-coil forward operator with
    8 coils, PnP-ADMM on top of it (the image-domain x-update of mc_kernels.hip) with the full-size network, and the dictionary match at K = 98 304 --
    one pipeline at 224 x 224, against the oracle's numpy restatement of the same loop; maps bit-exact for the same X."""
    T, N, nc = 1000, 224, 8
    dic = synth.make_dictionary(T=T, n_t1=384, n_t2=256)
    X0 = synth.synthesize_tsmi(synth.make_phantom_qmaps(N, seed=0), dic)
    fp, k = oracle.spiral_mask(N, 771, T)
    op = oracle.Operator(N, N, dic["V"], fp, k)
    hh, ww = np.meshgrid(np.linspace(-1, 1, N), np.linspace(-1, 1, N), indexing="ij")
    maps = np.stack([np.exp(-((hh - np.cos(a)) ** 2 + (ww - np.sin(a)) ** 2)) * np.exp(1j * (a + hh * ww)) for a in np.linspace(0, 2 * np.pi, nc, endpoint=False)], axis=2)
    maps = maps / np.sqrt(np.sum(np.abs(maps) ** 2, axis=2, keepdims=True))
    y_mc = np.stack([synth.awgn_measured(col, 30.0, seed=j) for j, col in enumerate(op.forward_mc(X0, maps).T)], axis=1)
    w = synth.structured_weights(seed=2, eps=0.3)
    e = engine_mod.Engine(0)
    e.set_operator(N, N, dic["V"], fp, k, max_batch=4)              # 8 coils through chunks of 4
    e.set_coils(maps)
    e.set_denoiser(w, N, N)
    e.set_dictionary(dic["D"], dic["normD"], dic["lut"])
    xg, lg = e.pnp_admm_mc(y_mc, iters=2)
    xo, lo = oracle.pnp_admm_mc(op, oracle.Net(w), y_mc, maps, iters=2)
    err = rel_err(xg, xo)
    print(f"cut0 x 8 coils: m {int(fp[-1])} per coil, lsqr gpu {lg.tolist()} oracle {lo.tolist()}, rel_err {err:.2e}")
    assert np.array_equal(lg, lo) and err < 1e-4
    mg = e.dict_match(xg)
    mx = oracle.dict_match(xg, dic["D"], dic["normD"], dic["lut"])
    assert np.array_equal(mg["dm"], mx["dm"]) and np.array_equal(mg["qmap"], mx["qmap"]) and np.array_equal(mg["pd"], mx["pd"])
    assert_atoms_close(mg, oracle.dict_match(xo, dic["D"], dic["normD"], dic["lut"]), dic, "cut0 x 8 coils")
    e.close()


def test_recon_batch_two_workers(engine_mod, oracle, synth, capfd):
    """qmri_recon_batch with two workers (host threads, one context each) -- both on device 0 here, one per GPU on a node: the
    slice shards are disjoint, every slice comes back in its own slot, results equal the single-worker run bit for bit."""
    from qmri_pnp_recon_poc_amd import batch
    dic, X0, fp, k, op, y0, nc, w = _small_case(oracle, synth)
    s = X0.shape[2]
    ys = np.stack([synth.awgn_measured(op.forward(synth.synthesize_tsmi(synth.make_phantom_qmaps(32, seed=sl), dic)), 30.0, seed=sl)
                   for sl in range(5)])
    kw = dict(N=32, M=32, V=dic["V"], frame_ptr=fp, kidx=k, weights=w, in_nc=s, out_nc=s, nc=nc, nb=2, dictionary=dic, iters=4)
    one = batch.recon_batch([0], ys, slices_per_launch=2, **kw)
    capfd.readouterr()
    two = batch.recon_batch([0, 0], ys, slices_per_launch=2, **kw)
    assert "timed out" not in capfd.readouterr().err                 # (workers that share a device do not start on the forms that need it to themselves)
    assert np.array_equal(one["X"], two["X"])
    assert np.array_equal(one["qmap"], two["qmap"])
    xo, _, _ = oracle.pnp_admm(op, oracle.Net(w, in_nc=s, out_nc=s, nc=nc, nb=2), ys[4], iters=4)
    assert rel_err(two["X"][4], xo) < 1e-4


def test_recon_batch_eight_workers_sixteen_slices(engine_mod, oracle, synth, capfd):
    """The N = 8 shape of qmri_recon_batch without eight GPUs: EIGHT workers -- eight host threads, eight contexts -- in ONE
    process, sixteen slices, two per launch, all on device 0 here (one per GPU on a node).  Every slice comes back in its own slot and equals the
    one-worker run bit for bit.  Workers that share a device start on the two-launch LSQR iteration and one launch per layer (the one-launch forms
    need the device to themselves; round 5's advice): nothing may time out on the way."""
    from qmri_pnp_recon_poc_amd import batch
    dic, X0, fp, k, op, y0, nc, w = _small_case(oracle, synth)
    s = X0.shape[2]
    ys = np.stack([synth.awgn_measured(op.forward(synth.synthesize_tsmi(synth.make_phantom_qmaps(32, seed=sl), dic)), 30.0, seed=sl)
                   for sl in range(16)])
    kw = dict(N=32, M=32, V=dic["V"], frame_ptr=fp, kidx=k, weights=w, in_nc=s, out_nc=s, nc=nc, nb=2, dictionary=dic, iters=3)
    one = batch.recon_batch([0], ys, slices_per_launch=2, **kw)
    capfd.readouterr()
    eight = batch.recon_batch([0] * 8, ys, slices_per_launch=2, **kw)
    err = capfd.readouterr().err
    assert "timed out" not in err, err
    assert np.array_equal(one["X"], eight["X"]) and np.array_equal(one["qmap"], eight["qmap"]) and np.array_equal(one["pd"], eight["pd"])
    assert len({one["X"][i].tobytes() for i in range(16)}) == 16    # sixteen different slices, each in its own slot
    xo, _, _ = oracle.pnp_admm(op, oracle.Net(w, in_nc=s, out_nc=s, nc=nc, nb=2), ys[15], iters=3)
    assert rel_err(eight["X"][15], xo) < 1e-4


@pytest.mark.parametrize("config", ["spiral_single_level_10ch", "epi_multi_level_11ch"])
def test_admm_224_full_length_100_iterations(engine_mod, oracle, synth, case224, config):
    """The reconstruction as the reference runs it (main_recon_tsmis_FFT.m:285-293: 100 ADMM iterations, gamma 0.05, lsqr
    1e-4 / 100) at the headline size, followed by the dictionary match (:300-317), GPU against the oracle end to end.
    Asserted: SURVEY.md section 8(d) metric 3 -- TSMI PSNR between the two reconstructions (:362-367 definition),
    fraction of pixels with the identical atom (T1/T2), PD relative error."""
    dic = case224["dic"]
    multi = config.startswith("epi")
    if multi:
        fp, k = oracle.epi_mask(224, 224, 1 / 65, 200)
        op = oracle.Operator(224, 224, dic["V"], fp, k)
        y = synth.awgn_measured(op.forward(case224["X0"]), 30.0, seed=3)
        w = synth.structured_weights(in_nc=11, out_nc=10, seed=5, eps=0.02)
    else:
        fp, k, op, y = case224["fp"], case224["k"], case224["op"], case224["y"]
        w = synth.structured_weights(seed=2, eps=0.02)
    e = engine_mod.Engine(0)
    e.set_operator(224, 224, dic["V"], fp, k)
    e.set_denoiser(w, 224, 224, in_nc=11 if multi else 10)
    e.set_dictionary(dic["D"], dic["normD"], dic["lut"])
    xg, _, lg = e.pnp_admm(y, iters=100, multi_level=multi, noise_std=0.01)
    mg = e.dict_match(xg)
    net = oracle.Net(w, in_nc=11 if multi else 10)
    xo, _, lo = oracle.pnp_admm(op, net, y, iters=100, multi_level=multi, noise_std=0.01)
    mo = oracle.dict_match(xo, dic["D"], dic["normD"], dic["lut"])
    err, psnr = rel_err(xg, xo), tsmi_psnr(xg, xo)
    same = float(np.mean(mg["dm"] == mo["dm"]))
    pd_err = float(np.linalg.norm(np.abs(mg["pd"]) - np.abs(mo["pd"])) / np.linalg.norm(np.abs(mo["pd"])))
    frac_counts, maxdiff = float(np.mean(lg == lo)), int(np.abs(lg - lo).max())
    print(f"{config}: rel_err {err:.3e} psnr_vs_oracle {psnr:.1f} dB, identical atoms {same:.5f}, pd rel err {pd_err:.2e}, "
          f"identical lsqr counts {frac_counts:.2f} (max diff {maxdiff}), lsqr total gpu {int(lg.sum())} oracle {int(lo.sum())}")
    # Tolerances: every x-update stops within ~2e-4 of the exact minimiser (LSQR tol 1e-4), the fp32
    # network differs in summation order: x after 100 iterations within 1e-3 relative, >= 70 dB between the two TSMIs,
    # >= 99 % of the pixels on the identical atom (differences sit on atoms of equal correlation to 1e-7), PD within 1e-3.
    assert err < 1e-3
    assert psnr > 70.0
    assert same > atom_tolerance(dic["K"])                              # K-aware (0.99 at this K = 8 192; see conftest.atom_tolerance)
    assert_atoms_close(mg, mo, dic, config)                             # ... and every differing pixel on a neighbouring atom of the grid
    assert pd_err < 1e-3
    assert frac_counts > 0.7 and maxdiff <= 2
    # same X in -> bit-exact maps out (the match itself is bit-exact; the differences above come from x)
    mx = oracle.dict_match(xg, dic["D"], dic["normD"], dic["lut"])
    assert np.array_equal(mx["dm"], mg["dm"]) and np.array_equal(mx["qmap"], mg["qmap"])
    e.close()


def test_fused_small_launches_equal_separate_kernels(tmp_path):
    """Round 4 folds the launches between the network and the solve into their neighbours (k_dual_fwd_h, k_ks_init_a<FWDW>, min / max in k_adj_h).
    The arithmetic of x, u, z is the same; only the partial sums of |z|^2 (the solve's stop threshold) are added in another fixed order.  With
    QMRI_DEBUG="fuse_ew=0" (two child processes) the separate kernels run: after 8 ADMM iterations at 224 x 224 -- single slice
    and a batch of 3, single- and multi-level denoiser -- the LSQR iteration counts must be identical and x equal to rounding (1e-12)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, numpy as np\n"
        f"sys.path.insert(0, {root!r})\n"
        "from qmri_pnp_recon_poc_amd import engine as E, synth, batch\n"
        "dic = synth.make_dictionary(T=200, n_t1=32, n_t2=16)\n"
        "fp, k = E.build_spiral(224, 771, 200)\n"
        "out = {}\n"
        "for multi in (0, 1):\n"
        "    w = synth.structured_weights(in_nc=10 + multi, seed=2, eps=0.3)\n"
        "    e = E.Engine(0)\n"
        "    e.set_operator(224, 224, dic['V'], fp, k, max_batch=3)\n"
        "    e.set_denoiser(w, 224, 224, in_nc=10 + multi, max_batch=3)\n"
        "    ys = []\n"
        "    for sd in range(3):\n"
        "        X0 = synth.synthesize_tsmi(synth.make_phantom_qmaps(224, seed=sd), dic)\n"
        "        ys.append(synth.awgn_measured(e.forward(X0), 30.0, seed=sd))\n"
        "    x, _, li = e.pnp_admm(ys[0], iters=8, multi_level=bool(multi))\n"
        "    out[f'x{multi}'] = x; out[f'li{multi}'] = li\n"
        "    e.close()\n"
        "    r = batch.recon_batch([0], np.stack(ys), N=224, M=224, V=dic['V'], frame_ptr=fp, kidx=k, weights=w, in_nc=10 + multi,\n"
        "                          iters=8, multi_level=bool(multi), slices_per_launch=3)\n"
        "    out[f'xb{multi}'] = r['X']\n"

        "np.savez(sys.argv[1], **out)\n")
    res = {}
    for flag in ("1", "0"):
        path = str(tmp_path / f"fuse_{flag}.npz")
        r = subprocess.run([sys.executable, "-c", code, path], env=dict(os.environ, QMRI_DEBUG="fuse_ew=" + flag), capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-3000:]
        res[flag] = np.load(path)
    for k in res["1"].files:
        a, b = res["1"][k], res["0"][k]
        if k.startswith("li"):
            assert np.array_equal(a, b), (k, a, b)
        else:
            assert rel_err(a, b) < 1e-12, (k, rel_err(a, b))


def test_health_counters_are_zero_on_a_clean_run_and_count_what_was_provoked(engine_mod, synth):
    """qmri_get_health (round 6): the fast paths that check themselves -- the one-launch LSQR iteration, the resident-tile convolution launch, the
    f16 scheme's range guards -- count every time they give up and the work is repeated on the slower path.  A clean reconstruction leaves every
    counter at zero and both forms armed; a provoked LSQR time-out and a provoked hand-off time-out each show up, with the same x."""
    E = engine_mod
    N, T, s, S = 64, 48, 10, 200
    dic = synth.make_dictionary(T=T, n_t1=16, n_t2=8, s=s)
    fp, k = E.build_spiral(N, S, T)
    w = synth.structured_weights(in_nc=s, out_nc=s, nc=(64, 64, 64, 64), nb=2, seed=5, eps=0.05)      # (nc[0] = 64: the resident-tile form applies)
    eng = E.Engine(0)
    eng.set_operator(N, N, dic["V"], fp, k)
    eng.set_denoiser(w, N, N, nc=(64, 64, 64, 64), nb=2)
    y = synth.awgn_measured(eng.forward(synth.synthesize_tsmi(synth.make_phantom_qmaps(N, seed=1), dic)), 30.0, seed=1)
    h = eng.health()
    assert h["denoiser_scheme"] == "f16x3" and h["set_denoiser_ms"]["pack_and_upload"] > 0 and h["set_denoiser_ms"]["calibration_probe"] > 0
    eng.profile_enable(3)                                           # stage marks: no wait inside the loop, resolved after the call
    x0, _, li0 = eng.pnp_admm(y, iters=4)
    eng.profile_enable(0)
    h = eng.health()
    assert (h["denoiser_fallbacks"], h["resident_tile_timeouts"], h["lsqr_one_launch_timeouts"], h["repeated_calls"]) == (0, 0, 0, 0)
    assert h["resident_tile_launch_armed"] and h["lsqr_one_launch"] == "armed" and h["last_call_wall_ms"] > 0
    st = h["last_call_stage_ms"]
    assert st["xupdate"] > 0 and st["denoiser"] > 0 and st["elementwise"] > 0 and st["diagnostics"] == 0
    assert sum(st.values()) <= h["last_call_wall_ms"] * 1.001
    # the marks change nothing, and a run without them leaves the stage fields at zero
    x1, _, li1 = eng.pnp_admm(y, iters=4)
    assert np.array_equal(x0, x1) and np.array_equal(li0, li1) and sum(eng.health()["last_call_stage_ms"].values()) == 0
    # provoked: one partial sum of the one-launch LSQR kernel is withheld -> time-out, the reconstruction is repeated on the two-launch iteration
    eng._check(eng.L.qmri_debug_lsqr_persist(eng.h, 2))
    x2, _, li2 = eng.pnp_admm(y, iters=4)
    h = eng.health()
    assert np.array_equal(x0, x2) and np.array_equal(li0, li2)
    assert h["lsqr_one_launch_timeouts"] == 1 and h["repeated_calls"] == 1 and h["lsqr_one_launch"] == "off"
    # provoked: one tile of the resident-tile launch publishes nothing -> its neighbours time out, the forward pass is repeated, one launch per layer
    eng._check(eng.L.qmri_debug_conv_resident(eng.h, 2, None))
    x3, _, _ = eng.pnp_admm(y, iters=2)
    h = eng.health()
    assert h["resident_tile_timeouts"] >= 1 and h["repeated_calls"] >= 2 and not h["resident_tile_launch_armed"] and h["denoiser_fallbacks"] == 0
    eng.close()
