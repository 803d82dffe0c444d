"""CPU: the MATLAB gateway (mex/qmri_mex.cpp) compiled, linked against libqmri.so and RUN under the mock MEX runtime (tests/cpp/mex_mock.cpp,
tests/mexmock.py) -- the commands that need no GPU: the mask builders, usage errors, and the loud failure of everything else without a device."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import mexmock  # noqa: E402
from mexmock import MexError, qmri_mex  # noqa: E402

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_gateway_links_against_libqmri_and_builds_the_masks_of_the_reference():
    """`[fp, k] = qmri_mex('build_spiral', N, S, T)` / `('build_epi', N, M, pct, T)` run on the host (no context): they must equal the fixtures
    of the independent restatement of setup_subsampling_*.m (tests/golden/matlab_rows_*.npz) bit for bit, as int32 column vectors."""
    for name, cmd, size in (("spiral", "build_spiral", 64), ("spiral", "build_spiral", 224), ("epi", "build_epi", 32), ("epi", "build_epi", 224)):
        g = np.load(os.path.join(GOLDEN, f"matlab_rows_{name}_{size}.npz"))
        N, T, prm = int(g["N"]), int(g["T"]), float(g["param"])        # (param: S of the spiral / the EPI sampling rate)
        args = (N, prm, T) if name == "spiral" else (N, N, prm, T)
        fp, k = qmri_mex(cmd, *[float(a) for a in args], nargout=2)
        assert fp.dtype == np.int32 and k.dtype == np.int32 and fp.shape == (T + 1, 1) and k.shape == (int(g["frame_ptr"][-1]), 1)
        assert np.array_equal(fp.ravel(), g["frame_ptr"]) and np.array_equal(k.ravel(), g["kidx"])


def test_gateway_usage_errors_are_matlab_errors():
    with pytest.raises(MexError) as e:
        qmri_mex("no_such_command")
    assert e.value.id == "qmri:usage" and "no_such_command" in e.value.msg
    with pytest.raises(MexError) as e:
        qmri_mex("build_spiral", 64.0)                              # too few arguments: a usage error, not a crash
    assert e.value.id == "qmri:usage"
    with pytest.raises(MexError) as e:
        qmri_mex("recon_batch", np.zeros((4, 2), np.complex128), {"iter": 1}, np.array([0.0]), 1.0, np.array([2.0, 2.0, 1.0]), nargout=1)
    assert e.value.id == "qmri:recon_batch:state"                  # nothing planned yet
    with pytest.raises(MexError) as e:
        qmri_mex("set_dictionary", np.zeros((8, 2), np.complex64), np.ones(8, np.float32), np.ones((8, 2), np.float32))
    assert e.value.id == "qmri:set_dictionary:type"                # complex atoms are refused on this route too
    # round 5: what an array must be and hold is checked by the gateway before the library reads through the plain pointers of the C ABI
    with pytest.raises(MexError) as e:
        qmri_mex("forward", np.zeros((8, 8, 2)), nargout=1)
    assert e.value.id == "qmri:state" and "set_operator" in e.value.msg          # no operator planned: nothing to size x against
    fp, k = qmri_mex("build_spiral", 32.0, 40.0, 6.0, nargout=2)
    V = np.linalg.qr(np.random.default_rng(0).standard_normal((6, 2)))[0]
    with pytest.raises(MexError) as e:
        qmri_mex("set_operator", 32.0, 32.0, V, fp.astype(np.float64), k)          # frame_ptr as doubles: would be read as garbage int32
    assert e.value.id == "qmri:set_operator:type"
    with pytest.raises(MexError) as e:
        qmri_mex("set_operator", 32.0, 32.0, V, fp[:-1].copy(), k)                 # T + 1 entries wanted
    assert e.value.id == "qmri:set_operator:size"
    with pytest.raises(MexError) as e:
        qmri_mex("set_operator", 32.0, 32.0, V, fp, k[:-3].copy())                 # kidx shorter than frame_ptr(end): would be read past its end
    assert e.value.id == "qmri:set_operator:size"
    with pytest.raises(MexError) as e:
        qmri_mex("dict_match", np.zeros((16, 2), np.complex128), 2.0, nargout=1)
    assert e.value.id == "qmri:state"                               # no dictionary
    # round 6 (advice): scalars are checked BEFORE they are cast to an integer type (a NaN or a negative double cast to int / size_t is undefined)
    for bad in (float("nan"), -1.0, 2.5, float("inf")):
        with pytest.raises(MexError) as e:
            qmri_mex("set_operator", bad, 32.0, V, fp, k)
        assert e.value.id == "qmri:set_operator:size"
        with pytest.raises(MexError) as e:
            qmri_mex("set_operator", 32.0, 32.0, V, fp, k, bad)                     # max_batch
        assert e.value.id == "qmri:set_operator:size"
        with pytest.raises(MexError) as e:
            qmri_mex("device", bad)
        assert e.value.id == "qmri:device"
    with pytest.raises(MexError) as e:
        qmri_mex("denoise", np.zeros((8, 8, 2)), 2.0, nargout=1)
    assert e.value.id == "qmri:state" and "set_denoiser" in e.value.msg          # nothing to size the output against: refused, not guessed


def test_gateway_fails_loudly_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    fp, k = qmri_mex("build_spiral", 32.0, 40.0, 6.0, nargout=2)
    V = np.linalg.qr(np.random.default_rng(0).standard_normal((6, 2)))[0]
    with pytest.raises(MexError) as e:
        qmri_mex("set_operator", 32.0, 32.0, V, fp, k)              # well-formed: the first command that needs the device
    assert e.value.id == "qmri:create" and "no CPU fallback" in e.value.msg
    mexmock.mex_exit()
