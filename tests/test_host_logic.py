"""CPU: host-side logic that needs no GPU -- C-ABI symbols, FFT codelets, synthetic generators, slice sharding."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_abi_exports_every_declared_symbol():
    from qmri_pnp_recon_poc_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "qmri.h")).read()
    declared = sorted(set(re.findall(r"\b(qmri_[a-z0-9_]+)\s*\(", hdr)))
    assert sorted(_lib.SYMBOLS) == declared, "ctypes symbol table is out of sync with include/qmri.h"
    L = _lib.lib()                                   # builds with hipcc if needed; loads without a GPU
    for s in declared:
        assert hasattr(L, s), f"libqmri.so does not export {s}"
    assert L.qmri_abi_version() == 1


def test_one_debug_entry_point_and_one_environment_variable():
    """Round 5: the library's A/B and diagnostic switches sit behind ONE entry point (qmri_debug_knob) and ONE environment variable
    (QMRI_DEBUG="name=value,..."): a known name is accepted, an unknown one refused with a message, a malformed QMRI_DEBUG entry is reported on
    stderr instead of being silently ignored, and no other getenv is left in csrc/."""
    from qmri_pnp_recon_poc_amd import _lib
    L = _lib.lib()
    assert L.qmri_debug_knob(b"conv_xcd", 1) == 0
    assert L.qmri_debug_knob(b"no_such_knob", 1) == -1 and b"no_such_knob" in L.qmri_last_error(None)
    code = ("import sys; sys.path.insert(0, %r)\nfrom qmri_pnp_recon_poc_amd import _lib\nL = _lib.lib()\nprint(L.qmri_debug_knob(b'res_delay', 24))\n" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, QMRI_DEBUG="res_delay=32,typo_knob=1,conv_xcd"), capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.strip() == "0", r.stderr
    assert "typo_knob=1" in r.stderr and "'conv_xcd'" in r.stderr                  # unknown name / no value: both reported
    csrc = os.path.join(ROOT, "qmri_pnp_recon_poc_amd", "csrc")
    uses = [(f, n + 1) for f in sorted(os.listdir(csrc)) if f.endswith((".hip", ".cpp", ".h"))
            for n, line in enumerate(open(os.path.join(csrc, f))) if re.search(r"\bgetenv\s*\(", line)]
    assert uses == [("api_core.cpp", uses[0][1])] and len(uses) == 1, uses


def test_no_gpu_fails_loudly():
    """Without a usable gfx950 device the product refuses to run (no CPU fallback)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from qmri_pnp_recon_poc_amd import engine
    with pytest.raises(engine.QmriError) as ei:
        engine.Engine(0)
    assert "no HIP device" in str(ei.value) or "gfx950" in str(ei.value) or "failed" in str(ei.value)


def test_host_mask_builders_match_oracle(oracle):
    """qmri_build_spiral / qmri_build_epi are host integer code: bit-exact against the oracle, no GPU needed."""
    from qmri_pnp_recon_poc_amd import engine
    for (N, S, T) in [(224, 771, 200), (32, 120, 24), (64, 50, 7)]:
        fp, k = engine.build_spiral(N, S, T)
        fo, ko = oracle.spiral_mask(N, S, T)
        assert np.array_equal(fp, fo) and np.array_equal(k, ko)
    for (N, M, pct, T) in [(224, 224, 1 / 65, 200), (32, 32, 1 / 8, 40), (8, 4, 1 / 3, 5)]:
        fp, k = engine.build_epi(N, M, pct, T)
        fo, ko = oracle.epi_mask(N, M, pct, T)
        assert np.array_equal(fp, fo) and np.array_equal(k, ko)


def test_fft_codelets_on_host(tmp_path):
    exe = tmp_path / "fft_codelets_test"
    subprocess.run(["g++", "-O2", "-I", os.path.join(ROOT, "qmri_pnp_recon_poc_amd", "csrc"),
                    os.path.join(ROOT, "tests", "cpp", "fft_codelets_test.cpp"), "-o", str(exe)], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout


def test_synth_generators_are_deterministic(synth):
    a = synth.splitmix64(1234567, 4)
    assert a.tolist() == [6457827717110365317, 3203168211198807973, 9817491932198370423, 4593380528125082431]
    dic = synth.make_dictionary(T=60, n_t1=12, n_t2=8)
    V = dic["V"]
    assert np.abs(V.T @ V - np.eye(10)).max() < 1e-12                  # orthonormal columns (a7 relies on it)
    assert np.allclose(np.linalg.norm(dic["D"], axis=1), 1.0, atol=1e-6)
    q = synth.make_phantom_qmaps(64, seed=3)
    X = synth.synthesize_tsmi(q, dic)
    assert X.shape == (64, 64, 10) and np.all(X[:, :, 0] >= 0)         # sign-aligned to channel 1 (main_synthesize_tsmis.m:97-98)
    assert np.all(X[q[:, :, 2] == 0] == 0)
    y = np.ones(1000, complex)
    n = synth.awgn_measured(y, 30.0, seed=5) - y
    assert abs(10 * np.log10(1.0 / np.mean(np.abs(n) ** 2)) - 30.0) < 0.5   # measured SNR


def test_shard_slices():
    from qmri_pnp_recon_poc_amd.batch import shard_slices
    for nsl, world in [(120, 8), (15, 4), (3, 8), (0, 2)]:
        parts = [shard_slices(nsl, world, r) for r in range(world)]
        assert sorted(sum(parts, [])) == list(range(nsl))
        assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
    assert shard_slices(120, 8, 3) == list(range(45, 60))


def test_mex_shim_compiles_against_stub_header():
    """SYNTAX / TYPE CHECK ONLY: mex/qmri_mex.cpp through g++ -fsyntax-only against tests/stubs/mex.h (a declarations-only
    stand-in written from MathWorks' published C Matrix API; the image has no MATLAB).  It proves the gateway's calls into
    include/qmri.h and into the mx* API are well-formed C++; it does not build, link or run a MEX file."""
    r = subprocess.run(["g++", "-fsyntax-only", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "tests", "stubs"),
                        "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "qmri_pnp_recon_poc_amd", "mex", "qmri_mex.cpp")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    src = open(os.path.join(ROOT, "qmri_pnp_recon_poc_amd", "mex", "qmri_mex.cpp")).read()
    assert 'c == "set_dictionary"' in src and "mxIsComplex(prhs[a])" in src        # a complex D is refused, not reinterpreted


def test_complex_dictionary_is_refused_not_truncated(tmp_path, synth):
    """dict.D / V complex-typed: a zero imaginary part is accepted, a non-zero one raises on every route (Engine.set_dictionary,
    Engine.set_operator, batch.recon_batch, harness.load_dictionary) -- never a silent np.asarray(..., float32) truncation."""
    import scipy.io
    from qmri_pnp_recon_poc_amd import engine, harness, batch
    dic = synth.make_dictionary(T=40, n_t1=6, n_t2=5)
    Dz = dic["D"].astype(np.complex64)                                  # complex-typed, imaginary part zero: fine
    assert np.array_equal(engine.real_dictionary_array(Dz, "dict.D", np.float32), dic["D"])
    Dc = Dz.copy(); Dc[3, 2] += 0.25j
    with pytest.raises(ValueError, match="non-zero imaginary"):
        engine.real_dictionary_array(Dc, "dict.D", np.float32)
    with pytest.raises(ValueError, match="non-zero imaginary"):
        engine.real_dictionary_array(dic["V"] + 1e-3j, "V", np.float64)
    # file route: dict.D stored complex in the .mat
    for D, ok in ((Dz, True), (Dc, False)):
        path = tmp_path / f"dict_{ok}.mat"
        scipy.io.savemat(path, {"dict": {"V": dic["V"] + 0j, "D": D, "normD": dic["normD"], "lut": dic["lut"]}})
        if ok:
            out = harness.load_dictionary(str(path))
            assert out["D"].dtype == np.float32 and np.array_equal(out["D"], dic["D"]) and out["V"].dtype == np.float64
        else:
            with pytest.raises(ValueError, match="non-zero imaginary"):
                harness.load_dictionary(str(path))
    # batch route validates before it touches a device
    fp, k = engine.build_spiral(32, 60, 40)
    with pytest.raises(ValueError, match="non-zero imaginary"):
        batch.recon_batch([0], np.zeros((1, int(fp[-1])), complex), N=32, M=32, V=dic["V"], frame_ptr=fp, kidx=k,
                          weights=np.zeros(4, np.float32), dictionary={"D": Dc, "normD": dic["normD"], "lut": dic["lut"]})


ASAN_SCRIPT = r"""
import sys, numpy as np
sys.path.insert(0, %r)
from oracle import oracle as O
from qmri_pnp_recon_poc_amd import synth
N, T, s, S = 32, 24, 6, 120
dic = synth.make_dictionary(T=T, n_t1=12, n_t2=8, s=s)
X0 = synth.synthesize_tsmi(synth.make_phantom_qmaps(N, seed=0), dic)
fp, k = O.spiral_mask(N, S, T); fe, ke = O.epi_mask(N, N, 1 / 8, T)
op = O.Operator(N, N, dic["V"], fp, k); ope = O.Operator(N, N, dic["V"], fe, ke)
y = synth.awgn_measured(op.forward(X0), 30.0, seed=0)
x = op.adjoint(y); ope.adjoint(ope.forward(X0))
op.lsqr(y, 0.9 * x, 0.05, 1e-4, 100, x); op.direct(y, 0.9 * x, 0.05)
for multi in (0, 1):
    nc = (8, 16, 16, 32)
    w = synth.structured_weights(in_nc=s + multi, out_nc=s, nc=nc, nb=2, seed=3, eps=0.05)
    net = O.Net(w, in_nc=s + multi, out_nc=s, nc=nc, nb=2)
    xo, d, li = O.pnp_admm(op, net, y, iters=3, multi_level=bool(multi), gt=X0, want_diag=True)
O.dict_match(xo, dic["D"], dic["normD"], dic["lut"], want_xfit=True)
O.synthesize_tsmi(synth.make_phantom_qmaps(N, seed=1), dic["D"], dic["normD"], dic["lut"])
O.fista_lrtv(op, y, K=1e-3, iters=3)
print("ASAN_RUN_OK", O.admm_stage_seconds()["denoiser"] > 0)
"""


def test_oracle_under_address_and_ub_sanitizer(tmp_path):
    """The oracle built with `make asan` (-fsanitize=address,undefined; GPU sanitizers are unavailable on the pool) driven through
    every entry point the suites use -- masks, operator, LSQR, closed form, network, ADMM (both denoiser types), dictionary match,
    synthesis, LRTV -- in a child process with libasan preloaded.  Any report fails the test."""
    odir = os.path.join(ROOT, "oracle")
    subprocess.run(["make", "-C", odir, "-s", "asan"], check=True)
    asan_rt = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    ubsan_rt = subprocess.run(["gcc", "-print-file-name=libubsan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(asan_rt) or not os.path.exists(asan_rt):
        pytest.skip("libasan runtime not found")
    script = tmp_path / "asan_run.py"
    script.write_text(ASAN_SCRIPT % ROOT)
    env = dict(os.environ, QMRI_ORACLE_LIB=os.path.join(odir, "_build", "liboracle_asan.so"),
               LD_PRELOAD=asan_rt + (":" + ubsan_rt if os.path.exists(ubsan_rt) else ""),
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:exitcode=77", UBSAN_OPTIONS="halt_on_error=1:exitcode=78:print_stacktrace=1",
               OMP_NUM_THREADS="4")
    r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "ASAN_RUN_OK True" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-4000:]


def test_product_host_code_under_address_and_ub_sanitizer(tmp_path):
    """The PRODUCT's host code under AddressSanitizer + UBSan: `make asan-host` compiles every source file of libqmri host-only (no device
    code, so it runs here) with -fsanitize=address,undefined, and tests/cpp/host_asan_driver.cpp drives the mask builders, the weight
    packers of every layer kind in both splitting schemes, qmri_net_nparams, the refusals of every entry point without a context or device,
    and the ONNX reader -- an untrusted file -- on a well-formed export and on ~150 truncated / bit-flipped / spliced ones.  Any report
    fails the test.  (Kernels are covered by the parity tests on the GPU; device sanitizers are not available on the pool.)"""
    import onnx_writer as ow
    from qmri_pnp_recon_poc_amd import synth
    csrc = os.path.join(ROOT, "qmri_pnp_recon_poc_amd", "csrc")
    subprocess.run(["make", "-C", csrc, "-s", "-j4", "asan-host"], check=True)
    rt_dirs = [d for d in sorted(os.listdir("/opt/rocm/lib/llvm/lib/clang")) if os.path.isdir(os.path.join("/opt/rocm/lib/llvm/lib/clang", d, "lib", "linux"))]
    if not rt_dirs:
        pytest.skip("clang sanitizer runtime not found")
    rt = os.path.join("/opt/rocm/lib/llvm/lib/clang", rt_dirs[-1], "lib", "linux")
    in_nc, out_nc, nc, nb = 10, 10, (8, 16, 16, 32), 2
    blob = synth.random_weights(in_nc=in_nc, out_nc=out_nc, nc=nc, nb=nb, seed=5)
    good = ow.unetres_model(ow.split_blob(blob, in_nc, out_nc, nc, nb), in_nc, out_nc, nc, nb)
    files = [tmp_path / "good.onnx"]
    files[0].write_bytes(good)
    rng = np.random.default_rng(4)
    n = len(good)
    variants = []
    for cut in sorted(set([0, 1, 2, 3, 7, 8, 15, 16, 31, 64, 100, n // 4, n // 2, n - 100, n - 9, n - 2, n - 1] + [int(v) for v in rng.integers(1, n, 40)])):
        variants.append(good[:cut])                                            # truncations, also inside varints and length prefixes
    header = min(n, 4096)                                                       # (field tags, lengths, dims live near the front and around the nodes)
    for _ in range(60):
        b = bytearray(good)
        for pos in rng.integers(0, header if rng.random() < 0.7 else n, int(rng.integers(1, 4))):
            b[int(pos)] ^= 1 << int(rng.integers(0, 8))
        variants.append(bytes(b))
    for _ in range(30):                                                         # lengths blown up, pieces spliced and repeated
        a, c = sorted(int(v) for v in rng.integers(0, n, 2))
        variants.append(good[:a] + bytes([0xFF, 0xFF, 0xFF, 0xFF, 0x0F]) + good[c:])
        variants.append(good[:c] + good[a:])
    for i, v in enumerate(variants):
        p = tmp_path / f"bad_{i:03d}.onnx"
        p.write_bytes(v)
        files.append(p)
    env = dict(os.environ, LD_LIBRARY_PATH=rt + ":" + os.environ.get("LD_LIBRARY_PATH", ""),
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:exitcode=77:allocator_may_return_null=1:max_allocation_size_mb=4096",
               UBSAN_OPTIONS="halt_on_error=1:exitcode=78:print_stacktrace=1")
    r = subprocess.run([os.path.join(csrc, "_build_asan", "host_asan_driver")] + [str(f) for f in files], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "HOST_ASAN_DRIVER_OK" in r.stdout, (r.returncode, r.stdout[-2000:], r.stderr[-6000:])
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-6000:]
    print(r.stdout.strip().splitlines()[-2:])
