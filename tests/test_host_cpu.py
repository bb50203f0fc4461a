"""CPU-only checks of the host side: C-ABI surface, module schema, error behaviour, schedule logic."""
import os
import subprocess

import numpy as np
import pytest
import torch

import conv_tasnet_amd as ctn
from conv_tasnet_amd import _lib
from conv_tasnet_amd.solver import _HalvingSchedule
from conftest import load_golden, ROOT
from oracle import ctn_oracle as O


def test_library_exports_every_symbol_the_header_declares():
    protos = _lib.parse_header()
    assert len(protos) >= 27
    out = subprocess.run(["nm", "-D", "--defined-only", ctn.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = {line.split()[-1] for line in out.splitlines() if " T " in line}
    missing = sorted(set(protos) - exported)
    assert not missing, missing
    ctn.lib.load()      # dlopen + bind every prototype; no compute


def test_host_side_entry_points_without_gpu():
    assert ctn.lib.ctn_version() >= 100
    assert ctn.lib.ctn_padded_frames(3199) == 3200 and ctn.lib.ctn_padded_frames(64) == 64
    assert ctn.lib.ctn_padded_frames(1) == 64
    assert ctn.lib.ctn_pw_stats_parts(8, 512, 3200) >= 100
    assert ctn.lib.ctn_pw_wgrad_workspace(8, 512, 256, 3200) % (512 * 256 * 4) == 0
    assert ctn.lib.ctn_dw_bwd_rows(3, 1) == 8 and ctn.lib.ctn_dw_bwd_rows(3, 0) == 3
    assert ctn.lib.ctn_sisnr_workspace(8, 2, 32000) == 8 * ctn.lib.ctn_sisnr_chunks(32000) * 12 * 8
    assert ctn.lib.ctn_last_error() is not None


def test_bad_arguments_return_error_codes_not_crashes():
    # null pointers / bad sizes are rejected before any launch, so this is safe without a GPU
    rc = ctn.lib.ctn_pw_gemm(0, 0, 0, 1, 4, 4, 4, 4, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0)
    assert rc == -1 and b"null" in ctn.lib.ctn_last_error()
    rc = ctn.lib.ctn_dw_fwd(16, 16, 16, 1, 4, 8, 8, 99, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0)
    assert rc == -1 and b"kernel size" in ctn.lib.ctn_last_error()
    with pytest.raises(ctn.CtnError):
        ctn.lib.call("ctn_im2col", 0, 0, 1, 100, 20, 20, 9, 64, 0)


@pytest.mark.parametrize("tag,nt,causal", [("gLN", "gLN", False), ("cLN", "cLN", True)])
def test_constructor_is_bitwise_the_reference_constructor(tag, nt, causal):
    g = load_golden("init_seed11_" + tag)
    N, L, B, H, P, X, R, C = [int(v) for v in g["cfg"]]
    torch.manual_seed(11)
    m = ctn.ConvTasNet(N, L, B, H, P, X, R, C, norm_type=nt, causal=causal)
    sd = m.state_dict()
    ref_keys = [k[2:] for k in g if k.startswith("p:")]
    assert list(sd.keys()) == ref_keys
    for k in ref_keys:
        assert np.array_equal(sd[k].numpy(), g["p:" + k]), k
    cfg = O.Config(N, L, B, H, P, X, R, C, norm_type=nt, causal=causal)
    assert [n for n, _ in m.named_parameters()] == list(O.param_shapes(cfg).keys())
    # D10: gamma/beta are xavier-initialised, PReLU slopes stay 0.25
    assert float(sd["separator.network.2.0.0.net.1.weight"]) == 0.25
    assert float(sd["separator.network.0.gamma"].std()) > 0.05


def test_paper_config_parameter_count():
    m = ctn.ConvTasNet(256, 20, 256, 512, 3, 8, 4, 2)
    ps = list(m.parameters())
    assert len(ps) == 294 and sum(p.numel() for p in ps) == 8710720


def test_serialize_and_load_round_trip(tmp_path):
    torch.manual_seed(0)
    m = ctn.ConvTasNet(16, 20, 8, 16, 3, 2, 1, 2, norm_type="cLN", causal=True, mask_nonlinear="softmax")
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    pkg = ctn.ConvTasNet.serialize(m, opt, 3, tr_loss=torch.zeros(4), cv_loss=torch.ones(4))
    assert sorted(pkg.keys()) == sorted(['N', 'L', 'B', 'H', 'P', 'X', 'R', 'C', 'norm_type', 'causal', 'mask_nonlinear',
                                         'state_dict', 'optim_dict', 'epoch', 'tr_loss', 'cv_loss'])
    g = load_golden("solver_traj")
    assert sorted(pkg.keys()) == sorted(str(k) for k in g["pkg_keys"])     # same package keys as the reference wrote
    path = str(tmp_path / "m.pth.tar")
    torch.save(pkg, path)
    m2 = ctn.ConvTasNet.load_model(path)
    assert m2.causal and m2.norm_type == "cLN" and m2.mask_nonlinear == "softmax"
    for (k1, v1), (k2, v2) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert k1 == k2 and torch.equal(v1, v2)


def test_reference_state_dict_loads():
    g = load_golden("model_tiny_cln_causal")
    N, L, B, H, P, X, R, C = [int(v) for v in g["cfg"]]
    m = ctn.ConvTasNet(N, L, B, H, P, X, R, C, norm_type="cLN", causal=True)
    m.load_state_dict({k[2:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("p:")})


def test_product_path_has_no_cpu_fallback():
    m = ctn.ConvTasNet(16, 20, 8, 16, 3, 2, 1, 2)
    with pytest.raises(ctn.CtnError):
        m(torch.randn(1, 400))
    with pytest.raises(ctn.CtnError):
        ctn.cal_loss(torch.randn(1, 2, 100), torch.randn(1, 2, 100), torch.tensor([100]))
    with pytest.raises(ValueError):
        ctn.ConvTasNet(16, 20, 8, 16, 3, 2, 1, 2, mask_nonlinear="sigmoid").separator.softmax_mask()


def test_missing_library_fails_loudly(monkeypatch):
    fresh = _lib._Lib()
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libctn_hip.so")
    with pytest.raises(ctn.CtnError, match="no CPU fallback"):
        fresh.load()


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "conv-tasnet_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert "oracle" not in src.replace("# oracle", ""), fn


def test_halving_schedule_matches_reference_rule():
    s = _HalvingSchedule(True, True)
    seq = [5.0, 4.0, 4.5, 4.6, 4.7, 4.8, 4.1, 4.2, 4.3, 4.4, 4.5, 4.6, 4.7, 4.8]
    out = [s.update(v) for v in seq]
    # misses: 0,0,1,2,3(halve),4(halve),0,1,2,3(halve),4,5,6,7(stop)
    assert [h for h, _ in out] == [False, False, False, False, True, True, False, False, False, True, True, True, True, True]
    assert [st for _, st in out][-1] is True and not any(st for _, st in out[:-1])
    off = _HalvingSchedule(False, True)
    assert all(off.update(v) == (False, False) for v in [3, 4, 5, 6, 7, 8, 9, 10])


def test_overlap_and_add_reference_main_example_via_oracle():
    g = load_golden("ola_main_int")
    out = O.overlap_and_add(torch.from_numpy(g["signal"]).double(), int(g["step"]))
    assert np.array_equal(out.numpy().astype(np.int64), g["result"])
