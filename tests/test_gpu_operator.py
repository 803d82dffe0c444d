"""GPU parity: forward / adjoint operator and the LSQR x-update through the C ABI vs the CPU oracle."""
import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng(engine_mod, case224):
    e = engine_mod.Engine(0)
    e.set_operator(224, 224, case224["dic"]["V"], case224["fp"], case224["k"], max_batch=1)
    yield e
    e.close()


def test_masks_match_oracle(engine_mod, oracle):
    fp, k = engine_mod.build_spiral(224, 771, 200)
    fo, ko = oracle.spiral_mask(224, 771, 200)
    assert np.array_equal(fp, fo) and np.array_equal(k, ko)
    fp, k = engine_mod.build_epi(224, 224, 1 / 65, 200)
    fo, ko = oracle.epi_mask(224, 224, 1 / 65, 200)
    assert np.array_equal(fp, fo) and np.array_equal(k, ko)


def test_forward_adjoint_vs_oracle(eng, case224):
    rng = np.random.default_rng(1)
    x = rng.standard_normal((224, 224, 10)) + 1j * rng.standard_normal((224, 224, 10))
    y = rng.standard_normal(eng.m) + 1j * rng.standard_normal(eng.m)
    op = case224["op"]
    assert rel_err(eng.forward(x), op.forward(x)) < 1e-12          # fp64 both sides; tolerance 1e-12 relative
    assert rel_err(eng.adjoint(y), op.adjoint(y)) < 1e-12
    # real-input entry (F.forward(double(X0)), main_recon_tsmis_FFT.m:237)
    assert rel_err(eng.forward(case224["X0"]), op.forward(case224["X0"])) < 1e-12
    # adjointness <Ax,y> = <x,A'y>
    lhs = np.vdot(y, eng.forward(x))
    rhs = np.vdot(eng.adjoint(y), x)
    assert abs(lhs - rhs) / abs(lhs) < 1e-12


def test_single_precision_boundary(eng, case224):
    """qmri_forward_f32 / qmri_adjoint_f32 (MATLAB `single` arrays at the boundary): computed in double, rounded once."""
    rng = np.random.default_rng(5)
    x = (rng.standard_normal((224, 224, 10)) + 1j * rng.standard_normal((224, 224, 10))).astype(np.complex64)
    y = (rng.standard_normal(eng.m) + 1j * rng.standard_normal(eng.m)).astype(np.complex64)
    op = case224["op"]
    yg, xg = eng.forward(x), eng.adjoint(y)
    assert yg.dtype == np.complex64 and xg.dtype == np.complex64
    assert np.array_equal(yg, op.forward(x.astype(np.complex128)).astype(np.complex64)) or rel_err(yg, op.forward(x.astype(np.complex128))) < 1e-7
    assert rel_err(xg, op.adjoint(y.astype(np.complex128))) < 1e-7            # one rounding to single
    xr = case224["X0"].astype(np.float32)                                      # real single input (F.forward(single(X0)))
    assert rel_err(eng.forward(xr), op.forward(xr.astype(np.float64))) < 1e-7


def test_epi_operator_vs_oracle(engine_mod, oracle, case224):
    fp, k = oracle.epi_mask(224, 224, 1 / 65, 200)
    V = case224["dic"]["V"]
    e = engine_mod.Engine(0)
    e.set_operator(224, 224, V, fp, k)
    op = oracle.Operator(224, 224, V, fp, k)
    rng = np.random.default_rng(2)
    x = rng.standard_normal((224, 224, 10)) + 1j * rng.standard_normal((224, 224, 10))
    y = rng.standard_normal(e.m) + 1j * rng.standard_normal(e.m)
    assert e.m == 134400
    assert rel_err(e.forward(x), op.forward(x)) < 1e-12
    assert rel_err(e.adjoint(y), op.adjoint(y)) < 1e-12
    e.close()


def test_lsqr_one_launch_equals_two_launch_iteration_bit_for_bit(engine_mod, oracle, case224):
    """k_ks_persist (all iterations of a solve in one launch; the partial sums of the two norms per iteration cross workgroups as tagged
    granules) against k_ks_a / k_ks_b (two launches per iteration): same arithmetic in the same order, so x, the iteration count and the
    flag must be IDENTICAL -- at tol 1e-4 (the reference's), at a tight tolerance (45 iterations) and at the maxit cap; also on a grid
    whose work units do not fill the chip (64 x 64) and with a warm start."""
    rng = np.random.default_rng(11)
    for N, S, T in ((224, 771, 200), (64, 200, 48)):
        if N == 224:
            V, fp, k, op, y = case224["dic"]["V"], case224["fp"], case224["k"], case224["op"], case224["y"]
        else:
            V = np.linalg.qr(rng.standard_normal((T, 10)))[0]
            fp, k = oracle.spiral_mask(N, S, T)
            op = oracle.Operator(N, N, V, fp, k)
            y = op.forward(rng.standard_normal((N, N, 10))) + 0.01 * (rng.standard_normal(int(fp[-1])) + 1j * rng.standard_normal(int(fp[-1])))
        e = engine_mod.Engine(0)
        e.set_operator(N, N, V, fp, k)
        x0 = op.adjoint(y)
        z = x0 + 0.1 * (rng.standard_normal(x0.shape) + 1j * rng.standard_normal(x0.shape))
        for tol, maxit, start in ((1e-4, 100, x0), (1e-10, 100, x0), (1e-12, 7, x0), (1e-4, 100, 0.5 * z), (1e-4, 1, x0)):
            e.lsqr_persist(True)
            xa, ita, fla = e.xupdate(y, z, 0.05, tol, maxit, start, solver="lsqr")
            e.lsqr_persist(False)
            xb, itb, flb = e.xupdate(y, z, 0.05, tol, maxit, start, solver="lsqr")
            assert (ita, fla) == (itb, flb), (N, tol, maxit, ita, itb, fla, flb)
            assert np.array_equal(xa, xb), (N, tol, maxit, rel_err(xa, xb))
            xo, ito, flo, _ = op.lsqr(y, z, 0.05, tol, maxit, start)
            assert (ita, fla) == (ito, flo) and rel_err(xa, xo) < 1e-10
        e.close()


def test_multi_coil_operator_extension(engine_mod, oracle):
    """BASELINE.json configs[4] names a "complex-valued multi-coil forward op".  The reference simulates a single coil (README.md:63), so this extension
    has NO reference counterpart and nothing pins it (parity unpinned): A_mc x = [A (C_j .* x)]_j, A_mc^H y = sum_j conj(C_j) .* A^H y_j on top of the
    parity-tested single-coil operator.  Checked: against the oracle's restatement (1e-12), exact adjointness <A x, y> = <x, A^H y>, the closed form for
    constant maps (A_mc x = [c_j A x]), one all-ones coil == the single-coil operator bit for bit, more coils than max_batch (chunks), cut0 (T = 1000)
    with 8 coils, and the error without maps."""
    rng = np.random.default_rng(21)
    for N, T, nc, maxb in ((32, 24, 5, 2), (224, 1000, 8, 4)):
        V = np.linalg.qr(rng.standard_normal((T, 10)))[0]
        fp, k = oracle.spiral_mask(N, 771 if N == 224 else 120, T)
        op = oracle.Operator(N, N, V, fp, k)
        e = engine_mod.Engine(0)
        e.set_operator(N, N, V, fp, k, max_batch=maxb)
        x = rng.standard_normal((N, N, 10)) + 1j * rng.standard_normal((N, N, 10))
        with pytest.raises(engine_mod.QmriError):
            e.forward_mc(x)                                           # no maps yet: QMRI_ERR_STATE
        hh, ww = np.meshgrid(np.linspace(-1, 1, N), np.linspace(-1, 1, N), indexing="ij")
        maps = np.stack([np.exp(-((hh - np.cos(a)) ** 2 + (ww - np.sin(a)) ** 2)) * np.exp(1j * (a + hh * ww)) for a in np.linspace(0, 2 * np.pi, nc, endpoint=False)], axis=2)
        e.set_coils(maps)
        y = e.forward_mc(x)
        assert y.shape == (e.m, nc) and rel_err(y, op.forward_mc(x, maps)) < 1e-12
        w = rng.standard_normal(y.shape) + 1j * rng.standard_normal(y.shape)
        xa = e.adjoint_mc(w)
        assert rel_err(xa, op.adjoint_mc(w, maps)) < 1e-12
        lhs, rhs = np.vdot(w.ravel(), y.ravel()), np.vdot(xa.ravel(), x.ravel())
        assert abs(lhs - rhs) / abs(lhs) < 1e-12                      # <A x, w> = <x, A^H w>
        cst = (rng.standard_normal(nc) + 1j * rng.standard_normal(nc))[None, None, :] * np.ones((N, N, 1))
        e.set_coils(cst)
        y1 = e.forward(x)
        assert rel_err(e.forward_mc(x), y1[:, None] * cst[0, 0][None, :]) < 1e-13
        e.set_coils(np.ones((N, N, 1)))
        assert np.array_equal(e.forward_mc(x)[:, 0], y1) and np.array_equal(e.adjoint_mc(y1[:, None]), e.adjoint(y1))
        e.set_coils(None)
        with pytest.raises(engine_mod.QmriError):
            e.adjoint_mc(w)
        e.close()


def _coil_maps(N, nc):
    hh, ww = np.meshgrid(np.linspace(-1, 1, N), np.linspace(-1, 1, N), indexing="ij")
    m = np.stack([np.exp(-((hh - np.cos(a)) ** 2 + (ww - np.sin(a)) ** 2)) * np.exp(1j * (a + hh * ww)) for a in np.linspace(0, 2 * np.pi, nc, endpoint=False)], axis=2)
    return m / np.sqrt(np.sum(np.abs(m) ** 2, axis=2, keepdims=True))          # sum_j |C_j|^2 = 1: the usual normalisation, ||A_mc|| <= 1


def test_multi_coil_x_update_and_reconstruction_extension(engine_mod, oracle, synth):
    """Round 6, BASELINE.json configs[4]: the reconstruction on top of the multi-coil operator -- the x-update of PnP_ADMM.m:102
    with [A_mc; sqrt(r) I] as an image-domain LSQR, and the loop of PnP_ADMM.m:76-146 around it.  The reference is single-coil (README.md:63): NO
    reference counterpart, parity unpinned; the checker is the oracle's numpy restatement (Operator.lsqr_mc / pnp_admm_mc: orc_lsqr.c's recurrences
    statement by statement on forward_mc / adjoint_mc).  Checked: x, iteration count and flag of the x-update against it (tol 1e-4 and a tight solve,
    warm start, maxit reached), the tight solve against the normal equations' residual, ONE all-ones coil == the single-coil x-update (count equal,
    x to 1e-10: same Krylov iteration in another domain), the full loop against the oracle's, more coils than max_batch, and the error paths."""
    rng = np.random.default_rng(33)
    N, T, s, S, nc = 32, 24, 6, 120, 5
    dic = synth.make_dictionary(T=T, n_t1=24, n_t2=16, s=s)
    fp, k = oracle.spiral_mask(N, S, T)
    op = oracle.Operator(N, N, dic["V"], fp, k)
    maps = _coil_maps(N, nc)
    X0 = synth.synthesize_tsmi(synth.make_phantom_qmaps(N, seed=0), dic)
    y_mc = op.forward_mc(X0, maps)
    y_mc = y_mc + 0.01 * np.abs(y_mc).mean() * (rng.standard_normal(y_mc.shape) + 1j * rng.standard_normal(y_mc.shape))
    z = X0 + 0.05 * (rng.standard_normal(X0.shape) + 1j * rng.standard_normal(X0.shape))
    e = engine_mod.Engine(0)
    e.set_operator(N, N, dic["V"], fp, k, max_batch=2)              # 5 coils through chunks of 2
    with pytest.raises(engine_mod.QmriError):
        e.xupdate_mc(y_mc, z, 0.05)                                 # no maps yet
    e.set_coils(maps)
    r = 0.05
    for tol, maxit, x0 in ((1e-4, 100, None), (1e-12, 200, None), (1e-4, 100, op.adjoint_mc(y_mc, maps)), (1e-12, 3, None)):
        xg, ig, fg = e.xupdate_mc(y_mc, z, r, tol=tol, maxit=maxit, x0=x0)
        xo, io, fo = op.lsqr_mc(y_mc, maps, z, r, tol=tol, maxit=maxit, x0=x0)
        assert (ig, fg) == (io, fo) and rel_err(xg, xo) < 1e-10, (tol, maxit, ig, io, fg, fo)
        if tol < 1e-10 and maxit > 100:                             # the minimiser: (A^H A + r I) x = A^H y + r z
            lhs = op.adjoint_mc(op.forward_mc(xg, maps), maps) + r * xg
            assert fg == 0 and rel_err(lhs, op.adjoint_mc(y_mc, maps) + r * z) < 1e-9
    # one all-ones coil: the multi-coil x-update IS the single-coil one (image-domain iteration here, k-space iteration there)
    e.set_coils(np.ones((N, N, 1)))
    y1 = y_mc[:, :1].copy()
    xm, im, fm = e.xupdate_mc(y1, z, r)
    x1, i1, f1 = e.xupdate(y1[:, 0], z, r)
    assert (im, fm) == (i1, f1) and rel_err(xm, x1) < 1e-10
    # the loop
    nch = (8, 16, 16, 32)
    w = synth.structured_weights(in_nc=s, out_nc=s, nc=nch, nb=2, seed=3, eps=0.05)
    e.set_coils(maps)
    with pytest.raises(engine_mod.QmriError):
        e.pnp_admm_mc(y_mc, iters=2)                                # no denoiser yet
    e.set_denoiser(w, N, N, in_nc=s, out_nc=s, nc=nch, nb=2)
    xg, lg = e.pnp_admm_mc(y_mc, iters=4)
    xo, lo = oracle.pnp_admm_mc(op, oracle.Net(w, in_nc=s, out_nc=s, nc=nch, nb=2), y_mc, maps, iters=4)
    assert np.array_equal(lg, lo) and rel_err(xg, xo) < 1e-4
    assert rel_err(xg, X0) < rel_err(op.adjoint_mc(y_mc, maps), X0)  # (and it is a reconstruction: closer to the truth than the adjoint it started from)
    e.close()


@pytest.mark.parametrize("mask,T", [("epi", 200), ("spiral", 1000), ("epi", 100)])
def test_lsqr_one_launch_with_the_other_unit_shapes_epi_and_cut0(engine_mod, oracle, mask, T):
    """Round 5: a single slice under an EPI mask (every k location sampled ~2.7 times: 784 units of 64 slots) or at cut0 (T = 1000: 56 samples per k,
    604 units of 1024 samples) now gets work units of 256 slots / 2560 samples, <= 250 of them, so the one-launch iteration applies there too
    (KS_CAPS in qmri_internal.h; the kernels are instantiated per shape).  Same arithmetic as the two-launch iteration on the same units: x, count
    and flag IDENTICAL; both equal to the oracle's count, x to 1e-10; a measurable difference in time is asserted nowhere -- tools/xupdate_times.py
    and the bench line's epi / cut0 objects report it."""
    rng = np.random.default_rng(5)
    N = 224
    V = np.linalg.qr(rng.standard_normal((T, 10)))[0]
    fp, k = oracle.epi_mask(N, N, 1 / 65, T) if mask == "epi" else oracle.spiral_mask(N, 771, T)
    op = oracle.Operator(N, N, V, fp, k)
    y = op.forward(rng.standard_normal((N, N, 10))) + 0.01 * (rng.standard_normal(int(fp[-1])) + 1j * rng.standard_normal(int(fp[-1])))
    e = engine_mod.Engine(0)
    e.set_operator(N, N, V, fp, k)
    x0 = op.adjoint(y)
    z = x0 + 0.1 * (rng.standard_normal(x0.shape) + 1j * rng.standard_normal(x0.shape))
    for tol, maxit, start in ((1e-4, 100, x0), (1e-9, 100, x0), (1e-4, 100, 0.5 * z)):
        e.lsqr_persist(True)
        xa, ita, fla = e.xupdate(y, z, 0.05, tol, maxit, start, solver="lsqr")
        e.lsqr_persist(False)
        xb, itb, flb = e.xupdate(y, z, 0.05, tol, maxit, start, solver="lsqr")
        assert (ita, fla) == (itb, flb), (mask, T, tol, ita, itb, fla, flb)
        assert np.array_equal(xa, xb), (mask, T, tol, rel_err(xa, xb))
        xo, ito, flo, _ = op.lsqr(y, z, 0.05, tol, maxit, start)
        assert (ita, fla) == (ito, flo) and rel_err(xa, xo) < 1e-10, (mask, T, tol, ita, ito)
    # the recovery path on these shapes: a withheld partial sum -> bounded waits, the message, the two-launch iteration, same result
    e.lsqr_persist(2)
    xc, itc, flc = e.xupdate(y, z, 0.05, 1e-4, 100, x0, solver="lsqr")
    e.lsqr_persist(False)
    xd, itd, fld = e.xupdate(y, z, 0.05, 1e-4, 100, x0, solver="lsqr")
    assert (itc, flc) == (itd, fld) and np.array_equal(xc, xd)
    e.close()


def test_lsqr_one_launch_time_out_falls_back_to_two_launch_iteration(engine_mod, oracle, case224, capfd):
    """The recovery path of k_ks_persist: with the test hook (mode 2) one workgroup withholds a partial sum, every waiting wave gives up after
    its bounded spin, nothing is stored, the library says so on stderr, repeats the solve with the two-launch iteration from the untouched
    inputs -- same x, count and flag -- and stays on the two-launch iteration afterwards."""
    op, y = case224["op"], case224["y"]
    e = engine_mod.Engine(0)
    e.set_operator(224, 224, case224["dic"]["V"], case224["fp"], case224["k"])
    x0 = op.adjoint(y)
    z = 0.9 * x0
    e.lsqr_persist(False)
    xb, itb, flb = e.xupdate(y, z, 0.05, 1e-4, 100, x0, solver="lsqr")
    e.lsqr_persist(2)
    xa, ita, fla = e.xupdate(y, z, 0.05, 1e-4, 100, x0, solver="lsqr")
    err = capfd.readouterr().err
    assert "timed out" in err
    assert (ita, fla) == (itb, flb) and np.array_equal(xa, xb)
    xc, itc, flc = e.xupdate(y, z, 0.05, 1e-4, 100, x0, solver="lsqr")     # (now on the two-launch iteration: no second message)
    assert "timed out" not in capfd.readouterr().err and np.array_equal(xc, xb)
    e.close()


def test_admm_recovers_from_a_one_launch_time_out(engine_mod, synth, case224, capfd):
    """The same recovery on the route the reconstruction takes: qmri_pnp_admm never waits inside its loop and reads the LSQR state of all
    iterations after its final synchronisation.  With the test hook the first x-update's kernel times out (every workgroup that gives up says
    so on the device, k_ks_final_w tells the host whatever workgroup 0 believed, and the later launches of the call return at once instead of
    spinning to their own time-outs); the call then repeats itself on the two-launch iteration: same x and LSQR counts as a run that never
    used the one-launch kernel, the message once, and a call that is not slow."""
    import time
    e = engine_mod.Engine(0)
    e.set_operator(224, 224, case224["dic"]["V"], case224["fp"], case224["k"])
    e.set_denoiser(synth.structured_weights(seed=2, eps=0.02), 224, 224)
    y = case224["y"]
    e.lsqr_persist(False)
    xb, _, lb = e.pnp_admm(y, iters=6)
    e.lsqr_persist(2)
    t0 = time.perf_counter()
    xa, _, la = e.pnp_admm(y, iters=6)
    dt = time.perf_counter() - t0
    err = capfd.readouterr().err
    assert err.count("timed out") == 1
    assert np.array_equal(la, lb) and np.array_equal(xa, xb)
    assert dt < 3.0, dt                                               # (one time-out of 0.1-0.3 s, not one per ADMM iteration)
    # maxit = 0 (accepted by the argument checks): no LSQR iteration, the final kernels still run, x = x0
    op = case224["op"]
    x0 = op.adjoint(y)
    for persist in (True, False):
        e.lsqr_persist(persist)
        x1, it1, fl1 = e.xupdate(y, 0.9 * x0, 0.05, 1e-4, 0, x0, solver="lsqr")
        assert it1 == 0 and fl1 == 1 and rel_err(x1, x0) < 1e-12
    e.close()


def test_lsqr_xupdate_vs_oracle(eng, case224):
    op, y = case224["op"], case224["y"]
    x0 = op.adjoint(y)
    z = x0.copy()
    xo, ito, flo, _ = op.lsqr(y, z, 0.05, 1e-4, 100, x0)
    xg, itg, flg = eng.xupdate(y, z, 0.05, 1e-4, 100, x0, solver="lsqr")
    assert (itg, flg) == (ito, flo)                                   # same iteration count and flag
    assert rel_err(xg, xo) < 1e-10
    # maxit cap: flag 1, iter == maxit
    xo2, ito2, flo2, _ = op.lsqr(y, z, 0.05, 1e-12, 5, x0)
    xg2, itg2, flg2 = eng.xupdate(y, z, 0.05, 1e-12, 5, x0, solver="lsqr")
    assert (itg2, flg2) == (ito2, flo2) == (5, 1)
    assert rel_err(xg2, xo2) < 1e-10


def test_direct_xupdate_vs_oracle(eng, case224):
    op, y = case224["op"], case224["y"]
    x0 = op.adjoint(y)
    z = 0.9 * x0
    xd = op.direct(y, z, 0.05)
    xg, _, _ = eng.xupdate(y, z, 0.05, solver="direct")
    assert rel_err(xg, xd) < 1e-10
    # and LSQR at tol 1e-4 lands within the stop-rule ambiguity of the exact minimiser
    xl, _, _ = eng.xupdate(y, z, 0.05, 1e-4, 100, x0, solver="lsqr")
    assert rel_err(xl, xd) < 5e-4


def test_lsqr_kspace_long_run_matches_closed_form(eng, case224):
    # many iterations: the scalar recurrences that carry the never-sampled k locations must stay consistent with the
    # vector iteration on the sampled ones; LSQR run to a tight tolerance must land on the exact minimiser
    op, y = case224["op"], case224["y"]
    x0 = 0.5 * op.adjoint(y)
    z = 0.9 * op.adjoint(y)
    xd = op.direct(y, z, 0.05)
    xg, itg, flg = eng.xupdate(y, z, 0.05, 1e-13, 100, x0, solver="lsqr")
    xo, ito, flo, _ = op.lsqr(y, z, 0.05, 1e-13, 100, x0)
    assert (itg, flg) == (ito, flo)
    assert rel_err(xg, xo) < 1e-9
    assert rel_err(xg, xd) < 1e-7


def test_lsqr_exact_start_and_zero_iterations(eng, case224):
    op, y = case224["op"], case224["y"]
    z = 0.9 * op.adjoint(y)
    xd = op.direct(y, z, 0.05)
    xg, itg, flg = eng.xupdate(y, z, 0.05, 1e-4, 100, xd, solver="lsqr")        # start at the minimiser: stops at once
    xo, ito, flo, _ = op.lsqr(y, z, 0.05, 1e-4, 100, xd)
    assert (itg, flg) == (ito, flo)
    assert rel_err(xg, xd) < 1e-9
    x0 = op.adjoint(y)
    xg, itg, flg = eng.xupdate(y, z, 0.05, 1e-4, 0, x0, solver="lsqr")          # maxit = 0 returns x0
    assert itg == 0 and rel_err(xg, x0) < 1e-12


def test_lsqr_epi_mask_vs_oracle(engine_mod, oracle, case224):
    # EPI: whole k-rows sampled in a few frames each -- a very different slot / work-unit structure from the spiral
    fp, k = oracle.epi_mask(224, 224, 1 / 65, 200)
    V = case224["dic"]["V"]
    e = engine_mod.Engine(0)
    e.set_operator(224, 224, V, fp, k)
    op = oracle.Operator(224, 224, V, fp, k)
    y = op.forward(case224["X0"])
    x0 = op.adjoint(y)
    z = 0.8 * x0
    xo, ito, flo, _ = op.lsqr(y, z, 0.05, 1e-4, 100, x0)
    xg, itg, flg = e.xupdate(y, z, 0.05, 1e-4, 100, x0, solver="lsqr")
    assert (itg, flg) == (ito, flo)
    assert rel_err(xg, xo) < 1e-10
    e.close()


@pytest.mark.parametrize("N,s,T,S", [(32, 6, 24, 120), (64, 10, 40, 300), (128, 8, 60, 500), (128, 3, 16, 200)])
def test_other_grid_sizes_operator_lsqr_and_lrtv(engine_mod, oracle, synth, N, s, T, S):
    """Every grid size the FFT kernels implement (32, 64, 128 beside 224), several channel counts: operator to 1e-12, the LSQR
    x-update with the oracle's iteration count, a few LRTV iterations with identical counts."""
    dic = synth.make_dictionary(T=T, n_t1=16, n_t2=12, s=s)
    fp, k = oracle.spiral_mask(N, S, T)
    fg, kg = engine_mod.build_spiral(N, S, T)
    assert np.array_equal(fp, fg) and np.array_equal(k, kg)
    op = oracle.Operator(N, N, dic["V"], fp, k)
    e = engine_mod.Engine(0)
    e.set_operator(N, N, dic["V"], fp, k)
    rng = np.random.default_rng(N + s)
    x = rng.standard_normal((N, N, s)) + 1j * rng.standard_normal((N, N, s))
    y = rng.standard_normal(e.m) + 1j * rng.standard_normal(e.m)
    assert rel_err(e.forward(x), op.forward(x)) < 1e-12
    assert rel_err(e.adjoint(y), op.adjoint(y)) < 1e-12
    yy = op.forward(x) + 0.01 * y
    z = op.adjoint(yy) * 0.9
    xg, ig, fg_ = e.xupdate(yy, z, 0.05, 1e-4, 100, x0=op.adjoint(yy))
    xo, io, fo_, _ = op.lsqr(yy, z, 0.05, tol=1e-4, maxit=100, x0=op.adjoint(yy))
    assert ig == io and fg_ == fo_
    assert rel_err(xg, xo) < 1e-10
    xl, il = e.lrtv(yy, K=1e-3, iters=5)
    xlo, ilo = oracle.fista_lrtv(op, yy, K=1e-3, iters=5)
    assert il["iters"] == ilo["iters"] and il["prox_iters_total"] == int(ilo["prox_iters"].sum()) and il["halvings"] == ilo["halvings"]
    assert rel_err(xl, xlo) < 1e-9
    e.close()


@pytest.mark.parametrize("mask", ["spiral", "epi"])
@pytest.mark.parametrize("T", [100, 300, 500, 1000])
def test_every_cut_operator_and_lsqr_224(engine_mod, oracle, synth, T, mask):
    """main_recon_tsmis_FFT.m:41-44 lists cut0..cut4 = T 1000/500/300/200/100 as a one-variable switch: every cut must run
    (T = 200 is the module fixture).  cut0's V (1000 x 10 doubles = 80 KB) sits in LDS next to the row spectra of the k-space
    LSQR kernels.  Operator to 1e-12 (fp64 both sides), LSQR x-update with the oracle's iteration count and flag, x to 1e-10."""
    N, s = 224, 10
    dic = synth.make_dictionary(T=T, n_t1=32, n_t2=16, s=s)
    if mask == "spiral":
        fp, k = oracle.spiral_mask(N, 771, T)
        fg, kg = engine_mod.build_spiral(N, 771, T)
    else:
        fp, k = oracle.epi_mask(N, N, 1 / 65, T)
        fg, kg = engine_mod.build_epi(N, N, 1 / 65, T)
        assert fp[-1] == 3 * 224 * T
    assert np.array_equal(fp, fg) and np.array_equal(k, kg)
    op = oracle.Operator(N, N, dic["V"], fp, k)
    e = engine_mod.Engine(0)
    e.set_operator(N, N, dic["V"], fp, k)
    assert e.m == op.m == fp[-1]
    rng = np.random.default_rng(T)
    x = rng.standard_normal((N, N, s)) + 1j * rng.standard_normal((N, N, s))
    y = rng.standard_normal(e.m) + 1j * rng.standard_normal(e.m)
    assert rel_err(e.forward(x), op.forward(x)) < 1e-12
    assert rel_err(e.adjoint(y), op.adjoint(y)) < 1e-12
    X0 = synth.synthesize_tsmi(synth.make_phantom_qmaps(N, seed=1), dic)
    yy = synth.awgn_measured(op.forward(X0), 30.0, seed=T)
    x0 = op.adjoint(yy)
    z = 0.9 * x0
    xo, io, fo, _ = op.lsqr(yy, z, 0.05, 1e-4, 100, x0)
    xg, ig, fg_ = e.xupdate(yy, z, 0.05, 1e-4, 100, x0, solver="lsqr")
    assert (ig, fg_) == (io, fo)
    assert rel_err(xg, xo) < 1e-10
    e.close()


def test_admm_two_contexts_on_one_device_concurrently(engine_mod, synth, case224, capfd):
    """Two reconstructions at once on one device, one context and one host thread each.  The one-launch LSQR kernel and the resident-tile
    convolution launch both need all their workgroups on the chip together; with a second context's kernels in between they may not get them:
    their waits are bounded, the calls repeat themselves on the forms that need no co-residency, and x and the LSQR counts come out identical
    to a run alone -- whatever the interleaving, and without the f16 range guard mistaking the neighbour's LDS leftovers for an overflow."""
    import threading
    y = case224["y"]
    w = synth.structured_weights(seed=2, eps=0.02)
    es = []
    for _ in range(2):
        e = engine_mod.Engine(0)
        e.set_operator(224, 224, case224["dic"]["V"], case224["fp"], case224["k"])
        e.set_denoiser(w, 224, 224)
        es.append(e)
    xr, _, lr = es[0].pnp_admm(y, iters=8)
    res = [None, None]

    def work(i):
        out = []
        for _ in range(3):
            out.append(es[i].pnp_admm(y, iters=8))
        res[i] = out

    ts = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=280)
    assert not any(t.is_alive() for t in ts)
    for i in range(2):
        for x, _, l in res[i]:
            assert np.array_equal(l, lr) and np.array_equal(x, xr)
        assert es[i].denoiser_scheme() == (2, 0)
    capfd.readouterr()
    for e in es:
        e.close()


def test_admm_same_bits_with_and_without_the_resident_tile_launch(engine_mod, synth, case224):
    """The reconstruction as a whole: eight PnP-ADMM iterations with the full-resolution ResBlocks (+ head, tail, down-sampling convolution) in the
    resident-tile launches and with one launch per layer -- identical x and LSQR counts (the network's output is the same bit for bit, so is
    everything behind it)."""
    e = engine_mod.Engine(0)
    e.set_operator(224, 224, case224["dic"]["V"], case224["fp"], case224["k"])
    e.set_denoiser(synth.structured_weights(seed=2, eps=0.02), 224, 224)
    y = case224["y"]
    e.conv_resident(0)
    x0, _, l0 = e.pnp_admm(y, iters=8)
    e.conv_resident(1)
    x1, _, l1 = e.pnp_admm(y, iters=8)
    assert np.array_equal(l0, l1) and np.array_equal(x0, x1)
    assert e.conv_resident(1) == 0 and e.denoiser_scheme() == (2, 0)
    e.close()
