"""GPU parity: dictionary match through the C ABI vs the CPU oracle (bit-exact atom indices)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("K", [(128, 64), (37, 5)])
def test_dict_match_bit_exact(engine_mod, oracle, synth, case224, K):
    dic = synth.make_dictionary(T=200, n_t1=K[0], n_t2=K[1])
    X = synth.synthesize_tsmi(case224["q"], dic)
    rng = np.random.default_rng(3)
    X = X + 0.01 * (rng.standard_normal(X.shape) + 1j * rng.standard_normal(X.shape))
    e = engine_mod.Engine(0)
    e.set_dictionary(dic["D"], dic["normD"], dic["lut"])
    g = e.dict_match(X)
    o = oracle.dict_match(X, dic["D"], dic["normD"], dic["lut"])
    assert np.array_equal(g["dm"], o["dm"])                       # index work: bit-exact
    assert np.array_equal(g["qmap"], o["qmap"])
    assert np.array_equal(g["mt"], o["mt"])
    assert np.array_equal(g["pd"], o["pd"])                         # same bits (no NaNs involved)
    e.close()


def test_dict_match_edge_cases(engine_mod, oracle, synth):
    dic = synth.make_dictionary(T=100, n_t1=9, n_t2=7)
    lut = dic["lut"].copy()
    lut[3, 1] = np.nan                                             # NaN in lut -> 0 (mrf_dtm_cpu.m:138)
    e = engine_mod.Engine(0)
    e.set_dictionary(dic["D"], dic["normD"], lut)
    X = np.zeros((5, 7, 10), np.complex128)                        # all-zero pixels: every |ip| ties at 0 -> first atom
    X[1, 1] = dic["D"][3] * 2.0                                    # exact atom 3 (1-based 4)
    X[2, 2] = -1j * dic["D"][10]
    g = e.dict_match(X)
    o = oracle.dict_match(X, dic["D"], dic["normD"], lut)
    assert np.array_equal(g["dm"], o["dm"])
    assert g["dm"][0, 0] == 1 and g["dm"][1, 1] == 4 and g["dm"][2, 2] == 11
    assert g["qmap"][1, 1, 1] == 0.0
    assert np.array_equal(g["pd"], o["pd"])                         # same bits (no NaNs involved)
    e.close()


def test_dict_match_magnitude_ties(engine_mod, oracle):
    """max(abs(ip)) compares single-precision magnitudes (mrf_dtm_cpu.m:92): atoms whose |ip|^2 differ in the last bits but whose
    abs is the same single tie, first index wins.  The constructed fixture (tools/gen_matlab_rows.py) and a stress case where
    thousands of atoms sit within a few ulps of the maximum -- the kernel's threshold bookkeeping against the oracle's plain
    sqrtf-per-candidate loop, bit for bit."""
    import os
    from conftest import GOLDEN
    g = np.load(os.path.join(GOLDEN, "matlab_rows_dictmatch.npz"))
    ones, lut = np.ones(6, np.float32), np.zeros((6, 2), np.float32)
    e = engine_mod.Engine(0)
    for D, want in ((g["Dt"], 2), (g["Dt2"], 1)):
        e.set_dictionary(D, ones, lut)
        r = e.dict_match(g["xt"])
        assert r["dm"][0] == want and r["mt"][0] == np.float32(1.0)
    # stress: K atoms = one direction + perturbations of a few ulps, in random order; 40 x 40 pixels along nearby directions
    rng = np.random.default_rng(9)
    K, s = 6000, 10
    base = rng.standard_normal(s).astype(np.float32)
    base /= np.linalg.norm(base)
    D = np.repeat(base[None, :], K, axis=0)
    bits = D.view(np.int32) + rng.integers(-3, 4, size=D.shape, dtype=np.int32)
    D = bits.view(np.float32).copy()
    nd = np.ones(K, np.float32)
    lut = np.stack([np.arange(K), np.arange(K)], axis=1).astype(np.float32)
    X = (base[None, None, :] * (1.0 + rng.random((40, 40, 1)))) * np.exp(1j * rng.random((40, 40, 1)) * 6.28)
    X = X + 1e-7 * rng.standard_normal(X.shape)
    e.set_dictionary(D, nd, lut)
    gq = e.dict_match(X)
    oq = oracle.dict_match(X, D, nd, lut)
    assert np.array_equal(gq["dm"], oq["dm"]) and np.array_equal(gq["mt"], oq["mt"]) and np.array_equal(gq["pd"], oq["pd"])
    # and the semantics matter on this input: an argmax over |ip|^2 picks a different atom for a good share of the pixels
    x32 = X.reshape(-1, s).astype(np.complex64)
    re = np.zeros((K, x32.shape[0]), np.float32); im = np.zeros_like(re)
    for c in range(s):
        re = (re + D[:, c:c + 1] * x32[:, c].real[None, :]).astype(np.float32)
        im = (im - D[:, c:c + 1] * x32[:, c].imag[None, :]).astype(np.float32)
    print("pixels where argmax |ip|^2 (numpy, unfused) differs from max(abs(ip)):",
          float(np.mean((np.argmax(re * re + im * im, axis=0) + 1) != gq["dm"].ravel())))
    e.close()


@pytest.mark.parametrize("K,npix_side", [(40000, 40), (32768, 224)], ids=["ties_across_atom_parts", "full_slice_four_parts"])
def test_dict_match_atoms_split_over_workgroups(engine_mod, oracle, synth, case224, K, npix_side):
    """Large dictionaries are matched in P parts of the atom range per pixel tile and merged by k_dict_merge (larger magnitude, then lower
    index): the same answer as one pass, bit for bit -- also when thousands of atoms ACROSS the parts tie within a few ulps."""
    rng = np.random.default_rng(21)
    s = 10
    e = engine_mod.Engine(0)
    if npix_side == 224:
        dic = synth.make_dictionary(T=200, n_t1=256, n_t2=128)
        assert dic["K"] == K
        D, nd, lut = dic["D"], dic["normD"], dic["lut"]
        X = synth.synthesize_tsmi(case224["q"], dic)
        X = X + 0.01 * (rng.standard_normal(X.shape) + 1j * rng.standard_normal(X.shape))
    else:
        base = rng.standard_normal(s).astype(np.float32)
        base /= np.linalg.norm(base)
        D = np.repeat(base[None, :], K, axis=0)
        D = (D.view(np.int32) + rng.integers(-3, 4, size=D.shape, dtype=np.int32)).view(np.float32).copy()
        nd = np.ones(K, np.float32)
        lut = np.stack([np.arange(K), np.arange(K)], axis=1).astype(np.float32)
        X = (base[None, None, :] * (1.0 + rng.random((npix_side, npix_side, 1)))) * np.exp(1j * rng.random((npix_side, npix_side, 1)) * 6.28)
        X = X + 1e-7 * rng.standard_normal(X.shape)
    e.set_dictionary(D, nd, lut)
    g = e.dict_match(X)
    o = oracle.dict_match(X, D, nd, lut)
    assert np.array_equal(g["dm"], o["dm"]) and np.array_equal(g["mt"], o["mt"]) and np.array_equal(g["pd"], o["pd"]) and np.array_equal(g["qmap"], o["qmap"])
    if npix_side != 224:
        print("distinct winning atoms:", len(np.unique(g["dm"])), "largest index", int(g["dm"].max()))
    e.close()


def test_dict_match_bench_K_98304_full_slice(engine_mod, oracle, synth, case224):
    """The dictionary size of the slices bench (384 x 256 grid, K = 98 304) on a full 224 x 224 slice: bit-exact maps against the
    oracle's blocked match (mrf_dtm_cpu.m:74: blockSize 1e9 -> 10 172 pixels per block at this K)."""
    dic = synth.make_dictionary(T=200, n_t1=384, n_t2=256)
    assert dic["K"] == 98304
    rng = np.random.default_rng(5)
    X = synth.synthesize_tsmi(case224["q"], dic)
    X = X + 0.005 * (rng.standard_normal(X.shape) + 1j * rng.standard_normal(X.shape))
    e = engine_mod.Engine(0)
    e.set_dictionary(dic["D"], dic["normD"], dic["lut"])
    g = e.dict_match(X)
    o = oracle.dict_match(X, dic["D"], dic["normD"], dic["lut"])
    assert np.array_equal(g["dm"], o["dm"]) and np.array_equal(g["mt"], o["mt"]) and np.array_equal(g["pd"], o["pd"]) and np.array_equal(g["qmap"], o["qmap"])
    assert len(np.unique(g["dm"])) > 500
    e.close()


def _same(a, b):
    return all(np.array_equal(a[k], b[k]) for k in ("dm", "mt", "pd", "qmap"))


def test_dict_match_f16_filter_changes_nothing_and_has_margin_to_spare(engine_mod, oracle, synth, case224):
    """The f16 filter only decides which 32-atom tiles get the exact single-precision products (dict_kernels.hip): with it and without it
    the maps are the oracle's, bit for bit -- on a noisy slice, on inputs built against the filter (channels over six decades, atoms with
    components down to 1e-6, pixels nearly orthogonal to every atom so that the winner is a small number among thousands of close
    ones) -- and the proven margin is far from the first wrong answer: the same inputs with the margin cut to 1/16 still agree."""
    rng = np.random.default_rng(77)
    e = engine_mod.Engine(0)
    cases = []
    dic = synth.make_dictionary(T=200, n_t1=96, n_t2=64)
    X = synth.synthesize_tsmi(case224["q"], dic)
    X = X + 0.02 * (rng.standard_normal(X.shape) + 1j * rng.standard_normal(X.shape))
    cases.append(("noisy slice", dic["D"], dic["normD"], dic["lut"], X[:96, :96]))
    K, s = 9000, 10
    D = rng.standard_normal((K, s)) * np.logspace(0, -6, s)[rng.permutation(s)][None, :]
    D = (D / np.linalg.norm(D, axis=1, keepdims=True)).astype(np.float32)
    lut = np.stack([np.arange(K), -np.arange(K)], axis=1).astype(np.float32)
    Xw = (rng.standard_normal((48, 48, s)) + 1j * rng.standard_normal((48, 48, s))) * np.logspace(2, -4, s)[None, None, :]
    cases.append(("six decades", D, np.ones(K, np.float32), lut, Xw))
    # pixels orthogonal to the dictionary's dominant directions up to 1e-4: |ip| << |x| |d|, thousands of atoms within the margin
    D2 = rng.standard_normal((K, s)); D2[:, 5:] *= 1e-4
    D2 = (D2 / np.linalg.norm(D2, axis=1, keepdims=True)).astype(np.float32)
    Xo = np.zeros((40, 40, s), np.complex128)
    Xo[..., 5:] = rng.standard_normal((40, 40, 5)) + 1j * rng.standard_normal((40, 40, 5))
    Xo[..., :5] = 1e-5 * rng.standard_normal((40, 40, 5))
    cases.append(("nearly orthogonal", D2, np.ones(K, np.float32), lut, Xo))
    s16 = 16
    D3 = rng.standard_normal((5000, s16)).astype(np.float32)
    D3 /= np.linalg.norm(D3, axis=1, keepdims=True)
    X3 = rng.standard_normal((32, 32, s16)) + 1j * rng.standard_normal((32, 32, s16))
    cases.append(("s = 16, scaled 1e12", D3 * np.float32(3.7), np.ones(5000, np.float32), lut[:5000], X3 * 1e12))
    for name, Dm, nd, lt, Xc in cases:
        o = oracle.dict_match(Xc, Dm, nd, lt)
        e.set_dictionary(Dm, nd, lt)
        e.dict_filter(False)
        g0 = e.dict_match(Xc)
        e.dict_filter(True)
        g1 = e.dict_match(Xc)
        assert _same(g0, o), name + ": exact products"
        assert _same(g1, o), name + ": with the filter"
        first_bad = None
        for k in range(1, 15):
            e.dict_filter(True, 2.0 ** -k)
            if not _same(e.dict_match(Xc), o):
                first_bad = k
                break
        e.dict_filter(True, 1.0)
        print(f"{name}: distinct atoms {len(np.unique(o['dm']))}; first wrong answer with the margin cut to 2^-{first_bad}")
        assert first_bad is None or first_bad > 4, name
    e.close()


def test_dict_match_filter_leaves_non_finite_and_extreme_pixels_to_the_exact_products(engine_mod, oracle, synth):
    """Pixels the filter cannot scale (huge, tiny, non-finite channels) go through the exact products tile by tile; zero pixels tie
    at 0 and keep the first atom."""
    dic = synth.make_dictionary(T=100, n_t1=40, n_t2=30)
    rng = np.random.default_rng(5)
    X = np.zeros((8, 8, 10), np.complex128)
    X[1] = dic["D"][rng.integers(0, dic["K"], 8)] * 1e-36
    X[2] = dic["D"][rng.integers(0, dic["K"], 8)] * 1e35
    X[3] = dic["D"][rng.integers(0, dic["K"], 8)] * (1 + 1j)
    X[4, 2, 3] = np.inf
    e = engine_mod.Engine(0)
    e.set_dictionary(dic["D"], dic["normD"], dic["lut"])
    g = e.dict_match(X)
    e.dict_filter(False)
    g0 = e.dict_match(X)
    o = oracle.dict_match(X, dic["D"], dic["normD"], dic["lut"])
    fin = np.ones((8, 8), bool); fin[4, 2] = False
    assert np.array_equal(g["dm"], g0["dm"])
    assert np.array_equal(g["dm"][fin], o["dm"][fin]) and np.array_equal(g["mt"][fin], o["mt"][fin])
    assert np.all(g["dm"][0] == 1)
    e.close()


def test_dict_match_filter_random_shapes(engine_mod, oracle):
    """Random channel counts (1 .. 16), dictionary sizes (not multiples of a tile or of an LDS step), pixel counts (not multiples of the
    128-pixel workgroup), magnitudes and phases: with and without the f16 filter the same bits, and every fourth case against the oracle."""
    rng = np.random.default_rng(2024)
    e = engine_mod.Engine(0)
    for case in range(32):
        s = int(rng.integers(1, 17))
        K = int(rng.choice([33, 97, 255, 256, 257, 1000, 2049, 5000]))
        npix = int(rng.choice([1, 31, 32, 127, 129, 300, 1000]))
        D = rng.standard_normal((K, s)).astype(np.float32)
        if case % 3 == 0:                                          # near-duplicate atoms: ties and near-ties across tiles
            D[K // 2:] = D[:K - K // 2] * (1 + 1e-7 * rng.standard_normal((K - K // 2, 1))).astype(np.float32)
        D /= np.maximum(np.linalg.norm(D, axis=1, keepdims=True), 1e-20)
        D = (D * np.float32(10.0 ** rng.integers(-3, 4))).astype(np.float32)
        nd = np.linalg.norm(D, axis=1).astype(np.float32)
        lut = rng.standard_normal((K, 2)).astype(np.float32)
        X = (rng.standard_normal((npix, s)) + 1j * rng.standard_normal((npix, s))) * 10.0 ** rng.integers(-6, 7)
        if case % 5 == 0:
            X[rng.integers(0, npix)] = 0
        if case % 2 == 0:                                          # pixels that ARE atoms (scaled, rotated): exact matches among near-duplicates
            idx = rng.integers(0, K, npix)
            X = D[idx].astype(np.complex128) * np.exp(1j * rng.random((npix, 1)) * 6.28) * 10.0 ** rng.integers(-3, 4)
        e.set_dictionary(D, nd, lut)
        e.dict_filter(True)
        g1 = e.dict_match(X)
        e.dict_filter(False)
        g0 = e.dict_match(X)
        assert _same(g0, g1), f"case {case}: s={s} K={K} npix={npix}"
        if case % 4 == 0:
            assert _same(g1, oracle.dict_match(X, D, nd, lut)), f"case {case} vs oracle: s={s} K={K} npix={npix}"
    e.dict_filter(True)
    e.close()


# ---------------------------------------------------------------------------------------------------------------------------------
# wide dictionaries: 16 < s <= 1024 channels (uncompressed fingerprints, s = T; mrf_dtm_cpu.m:41-50 takes T from size(data.X)) -- the
# channel-blocked f32-MFMA GEMM of dictw_kernels.hip against the oracle's sequential fmaf chain, bit for bit
# ---------------------------------------------------------------------------------------------------------------------------------
def _assert_same(g, o, keys=("dm", "mt", "pd", "qmap")):
    for k in keys:
        assert np.array_equal(g[k], o[k]), f"{k} differs: {int(np.sum(g[k] != o[k]))} of {g[k].size}"


@pytest.mark.parametrize("T", [17, 64, 200, 1000])
def test_wide_dict_match_bit_exact(engine_mod, oracle, synth, T):
    """K = 4096 atoms x T channels against 64 x 64 noisy complex pixels (some of them all-zero, some an exact atom): indices, magnitudes, PD
    and maps equal the oracle's bits.  T = 17 and 1000 are not multiples of the 16-channel stage (zero padding adds fma(0, 0, acc))."""
    dic = synth.make_dictionary(T=T, n_t1=64, n_t2=64, uncompressed=True)
    assert dic["D"].shape == (4096, T)
    q = synth.make_phantom_qmaps(64, seed=1)
    X = synth.synthesize_tsmi(q, dic)
    rng = np.random.default_rng(T)
    X = X * np.exp(1j * rng.random((64, 64, 1)) * 6.28) + 0.02 * X.std() * (rng.standard_normal(X.shape) + 1j * rng.standard_normal(X.shape))
    X[0, :8] = 0.0                                                  # all-zero pixels: every |ip| ties at 0 -> atom 1
    X[5, 5] = 3.0 * dic["D"][1234]
    X[6, 6] = -2.0j * dic["D"][4095]
    e = engine_mod.Engine(0)
    e.set_dictionary(dic["D"], dic["normD"], dic["lut"])
    g = e.dict_match(X, want_xfit=True)
    o = oracle.dict_match(X, dic["D"], dic["normD"], dic["lut"], want_xfit=True)
    _assert_same(g, o, ("dm", "mt", "pd", "qmap", "Xfit"))
    assert g["dm"][0, 0] == 1 and g["dm"][5, 5] == 1235 and g["dm"][6, 6] == 4096
    assert len(np.unique(g["dm"])) > 50                            # (the match is not degenerate)
    e.close()


def test_wide_dict_match_ragged_sizes_and_ties(engine_mod, oracle):
    """Atom and pixel counts that are not multiples of the 128 x 128 tile, an odd channel count, and thousands of atoms within +-3 ulp of
    each other spread over several atom parts: max(abs(ip)) with the first index winning (mrf_dtm_cpu.m:92) survives the split and the merge."""
    rng = np.random.default_rng(77)
    e = engine_mod.Engine(0)
    for K, s, shape in ((185, 23, (5, 7)), (6000, 40, (40, 40)), (1300, 129, (31, 9))):
        base = rng.standard_normal(s).astype(np.float32)
        base /= np.linalg.norm(base)
        D = np.repeat(base[None, :], K, axis=0)
        D = (D.view(np.int32) + rng.integers(-3, 4, size=D.shape, dtype=np.int32)).view(np.float32).copy()
        nd = (1.0 + rng.random(K)).astype(np.float32)
        lut = np.stack([np.arange(K), -np.arange(K)], axis=1).astype(np.float32)
        lut[K // 2, 1] = np.nan                                    # NaN -> 0 (mrf_dtm_cpu.m:138)
        X = (base[None, None, :] * (1.0 + rng.random(shape + (1,)))) * np.exp(1j * rng.random(shape + (1,)) * 6.28)
        X = X + 1e-7 * rng.standard_normal(X.shape)
        X[0, 0] = 0.0
        e.set_dictionary(D, nd, lut)
        g = e.dict_match(X, want_xfit=True)
        o = oracle.dict_match(X, D, nd, lut, want_xfit=True)
        _assert_same(g, o, ("dm", "mt", "pd", "qmap", "Xfit"))
        assert g["dm"][0, 0] == 1
        print(f"K={K} s={s}: distinct winning atoms {len(np.unique(g['dm']))}, largest index {int(g['dm'].max())}")
    e.close()


def test_wide_dict_match_full_slice_round_trip(engine_mod, synth):
    """BASELINE configs[4] at its pixel count: 224 x 224 pixels x T = 1000 uncompressed channels against K = 4096 atoms (size-independent
    property: maps -> TSMI -> match returns the atoms the synthesis used, PD included), and the device entry point."""
    T = 1000
    dic = synth.make_dictionary(T=T, n_t1=64, n_t2=64, uncompressed=True)
    q = synth.make_phantom_qmaps(224, seed=0)
    X = synth.synthesize_tsmi(q, dic)
    t1g, t2g = dic["t1_grid"], dic["t2_grid"]
    e = engine_mod.Engine(0)
    e.set_dictionary(dic["D"], dic["normD"], dic["lut"])
    g = e.dict_match(X)
    fg = q[:, :, 2] > 0
    want = dic["lut"][g["dm"] - 1]
    assert np.array_equal(g["qmap"][fg], want[fg])
    # the matched atom is the nearest grid atom of the phantom's (T1, T2), as the synthesis chose it
    i1 = np.abs(np.log(t1g)[None, :] - np.log(np.maximum(q[:, :, 0][fg], 1e-9))[:, None]).argmin(1)
    near = np.abs(g["qmap"][:, :, 0][fg] - t1g[i1].astype(np.float32))
    assert np.mean(near < 1e-6) > 0.98                             # (nearest in linear, not log, distance differs for a few)
    # PD: a single-precision chain of 1000 fmaf (<= 1000 x 2^-24 relative); where a neighbouring, nearly parallel atom wins within that
    # rounding, PD moves by the ratio of the two normD
    rel = np.abs(np.abs(g["pd"][fg]) - q[:, :, 2][fg]) / q[:, :, 2][fg]
    assert np.mean(rel < 2e-4) > 0.98 and np.median(rel) < 2e-5
    # device-resident entry point (buffers from the HIP runtime libqmri itself uses): same bits as the host-buffer call
    import ctypes as C
    path = next(l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l)
    hip = C.CDLL(path)
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipFree.argtypes = [C.c_void_p]
    npix = 224 * 224
    xb = np.ascontiguousarray(X.astype(np.complex128).reshape((npix, T), order="F").ravel(order="F"))
    dm, pd = np.empty(npix, np.int32), np.empty(2 * npix, np.float32)
    d_x, d_dm, d_pd = C.c_void_p(), C.c_void_p(), C.c_void_p()
    assert hip.hipMalloc(C.byref(d_x), xb.nbytes) == 0 and hip.hipMalloc(C.byref(d_dm), dm.nbytes) == 0 and hip.hipMalloc(C.byref(d_pd), pd.nbytes) == 0
    assert hip.hipMemcpy(d_x, xb.ctypes.data_as(C.c_void_p), xb.nbytes, 1) == 0
    e.dict_match_dev(d_x.value, npix, d_pd=d_pd.value, d_dm=d_dm.value)
    e.synchronize()
    assert hip.hipMemcpy(dm.ctypes.data_as(C.c_void_p), d_dm, dm.nbytes, 2) == 0 and hip.hipMemcpy(pd.ctypes.data_as(C.c_void_p), d_pd, pd.nbytes, 2) == 0
    assert np.array_equal(dm.reshape((224, 224), order="F"), g["dm"])
    assert np.array_equal(pd.view(np.complex64).reshape((224, 224), order="F"), g["pd"])
    for d in (d_x, d_dm, d_pd):
        hip.hipFree(d)
    e.close()


def test_xfit_narrow_dictionary(engine_mod, oracle, synth, case224):
    """out.Xfit = ip(dm) .* D(dm,:) (par.f.Xout, mrf_dtm_cpu.m:95,129-134) for the compressed s = 10 dictionary, filter on and off,
    atoms split over workgroups and not."""
    rng = np.random.default_rng(5)
    e = engine_mod.Engine(0)
    for n1, n2 in ((37, 5), (256, 128)):
        dic = synth.make_dictionary(T=200, n_t1=n1, n_t2=n2)
        X = synth.synthesize_tsmi(case224["q"], dic)
        X = X + 0.01 * (rng.standard_normal(X.shape) + 1j * rng.standard_normal(X.shape))
        e.set_dictionary(dic["D"], dic["normD"], dic["lut"])
        o = oracle.dict_match(X, dic["D"], dic["normD"], dic["lut"], want_xfit=True)
        for filt in (True, False):
            e.dict_filter(filt)
            g = e.dict_match(X, want_xfit=True)
            _assert_same(g, o, ("dm", "mt", "pd", "qmap", "Xfit"))
    e.dict_filter(True)
    e.close()


def test_wide_dict_match_random_shapes(engine_mod, oracle):
    """The channel-blocked match on 24 random problem shapes -- channel counts 17 .. 300 around the 16-channel stage, atom counts around the
    128-atom tile and the 16-part super-tile, pixel counts around the 128-pixel tile; unit-norm random atoms plus near-duplicates (atoms a few ulps
    apart, so that ties cross tiles and parts), pixels = scaled atoms + noise, some exactly zero, some a single non-zero channel: every output equal
    to the oracle's bits, Xfit included."""
    rng = np.random.default_rng(2024)
    e = engine_mod.Engine(0)
    for trial in range(24):
        s = int(rng.choice([17, 18, 31, 32, 33, 47, 48, 49, 64, 100, 129, 300]))
        K = int(rng.choice([1, 5, 31, 32, 33, 127, 128, 129, 255, 257, 1000, 2049, 5000]))
        npix = int(rng.choice([1, 31, 33, 127, 128, 129, 500, 1025]))
        D = rng.standard_normal((K, s)).astype(np.float32)
        D /= np.linalg.norm(D, axis=1, keepdims=True)
        ndup = min(K // 2, 40)
        if ndup:                                                   # near-duplicates of random atoms at random places
            src, dst = rng.integers(0, K, ndup), rng.integers(0, K, ndup)
            D[dst] = (D[src].view(np.int32) + rng.integers(-2, 3, size=(ndup, s), dtype=np.int32)).view(np.float32)
        nd = (0.5 + rng.random(K)).astype(np.float32)
        lut = rng.random((K, 2)).astype(np.float32)
        X = D[rng.integers(0, K, npix)] * (0.1 + rng.random((npix, 1))) * np.exp(1j * rng.random((npix, 1)) * 6.28)
        X = X + 0.05 * (rng.standard_normal(X.shape) + 1j * rng.standard_normal(X.shape)) / np.sqrt(s)
        X[rng.integers(0, npix, max(1, npix // 20))] = 0.0
        one = rng.integers(0, npix)
        X[one] = 0.0
        X[one, rng.integers(0, s)] = 1.0 - 2.0j
        e.set_dictionary(D, nd, lut)
        g = e.dict_match(X, want_xfit=True)
        o = oracle.dict_match(X, D, nd, lut, want_xfit=True)
        for k in ("dm", "mt", "pd", "qmap", "Xfit"):
            assert np.array_equal(g[k], o[k]), (trial, s, K, npix, k, int(np.sum(g[k] != o[k])))
    e.close()


def test_wide_dict_match_at_bench_size_K_98304_T_1000(engine_mod, oracle, synth):
    """BASELINE configs[4] at full size: one 224 x 224 slice of T = 1000 uncompressed channels against K = 98 304 atoms (19.7 TFLOP, ~0.15 s on
    the GPU).  The oracle needs minutes per slice, so it is run on 768 pixels spread over the slice -- every pixel is matched independently, and
    on those pixels atom index, magnitude, PD and the maps must equal the oracle's bits; on the whole slice the size-independent properties:
    the match of a noise-free pixel is an atom whose (T1, T2) sit within a grid step of the phantom's, and dm stays inside 1 .. K."""
    T = 1000
    dic = synth.make_dictionary(T=T, n_t1=384, n_t2=256, uncompressed=True)
    K = int(dic["K"])
    assert K == 98304 and dic["D"].shape == (K, T)
    q = synth.make_phantom_qmaps(224, seed=0)
    X = synth.synthesize_tsmi(q, dic).astype(np.complex128)
    rng = np.random.default_rng(1)
    X[:, :112] += 0.02 * X.real.std() * (rng.standard_normal((224, 112, T)) + 1j * rng.standard_normal((224, 112, T)))     # left half noisy, right half clean
    X *= np.exp(0.7j)
    e = engine_mod.Engine(0)
    e.set_dictionary(dic["D"], dic["normD"], dic["lut"])
    g = e.dict_match(X)
    e.close()
    assert g["dm"].min() >= 1 and g["dm"].max() <= K
    npix = 224 * 224
    sel = np.linspace(0, npix - 1, 768).astype(np.int64)
    xs = X.reshape(npix, T, order="F")[sel]
    o = oracle.dict_match(xs, dic["D"], dic["normD"], dic["lut"])
    for k in ("dm", "mt", "pd"):
        assert np.array_equal(g[k].ravel(order="F")[sel], o[k]), k
    assert np.array_equal(g["qmap"].reshape(npix, -1, order="F")[sel], o["qmap"])
    # clean half: the matched atom is (within the rounding of a 1000-term single-precision chain) the synthesis's nearest grid atom
    fg = (q[:, :, 2] > 0)
    fg[:, :112] = False
    t1g, t2g = dic["t1_grid"], dic["t2_grid"]
    i1 = np.searchsorted(t1g, g["qmap"][:, :, 0][fg] * (1 - 1e-6)); i1 = np.clip(i1, 0, t1g.size - 1)
    want1 = np.abs(t1g[None, :] - q[:, :, 0][fg][:, None]).argmin(1)
    assert np.mean(np.abs(i1 - want1) <= 1) > 0.98
