"""CPU: the device ISA of the dominant kernel, regenerated from the current sources (`make asm`, hipcc cross-compiles
gfx950 without a GPU), keeps the properties DESIGN.md 4.3 relies on.  The generated .s / resource_usage.txt are build
products (git-ignored), so what is checked is always what the shipped library was built from."""
import os
import re
import subprocess

import pytest

from conftest import ROOT

CSRC = os.path.join(ROOT, "ndarray-interp_amd", "csrc")
# eval_bucketed_kernel<double, ST_CUBIC, U = 8, CQ = 128, NT = true, FULL = true>: the Target's evaluation kernel
DOMINANT = "_ZN3ndi20eval_bucketed_kernelIdLi1ELi8ELi128ELb1ELb1EEEvNS_9Eval1ArgsIT_EE"
GATHER = "_ZN3ndi16eval_rows_kernelIdLi1ELi1ELb1EEEvNS_9Eval1ArgsIT_EE"


@pytest.fixture(scope="module")
def asm():
    srcs = [os.path.join(CSRC, f) for f in ("ndinterp_api.hip", "kernels.hpp", "host_logic.hpp", "common.hpp")]
    out = os.path.join(CSRC, "ndinterp_api.gfx950.s")
    if not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(p) for p in srcs):
        subprocess.run(["make", "-C", CSRC, "asm"], check=True, capture_output=True, timeout=600)
    return open(out).read(), open(os.path.join(CSRC, "resource_usage.txt")).read()


def _body(text, symbol):
    m = re.search(r"^" + re.escape(symbol) + r":.*?^\.Lfunc_end\d+:", text, flags=re.S | re.M)
    assert m, f"{symbol} not found in the generated assembly"
    return m.group(0)


def _resources(usage, symbol):
    m = re.search(r"Function Name: " + re.escape(symbol) + r" .*?LDS Size \[bytes/block\]: (\d+)", usage, flags=re.S)
    assert m, symbol
    block = m.group(0)
    get = lambda key: int(re.search(re.escape(key) + r": (\d+)", block).group(1))
    return {"vgprs": get("VGPRs"), "scratch": get("ScratchSize [bytes/lane]"), "occupancy": get("Occupancy [waves/SIMD]"),
            "vgpr_spill": get("VGPRs Spill"), "lds": int(m.group(1))}


def test_dominant_kernel_isa(asm):
    text, usage = asm
    body = _body(text, DOMINANT)
    # bit-exactness: the polynomial is evaluated in the reference's operation order, never contracted to FMA
    # (cubic_spline.rs:824-828; -ffp-contract=off)
    assert "v_fma_f64" not in body and "v_fmac_f64" not in body
    assert len(re.findall(r"\bv_mul_f64\b", body)) >= 64 and len(re.findall(r"\bv_add_f64\b", body)) >= 32
    # the output stream: 16-byte non-temporal stores, 8 per thread and query (U = 8 row segments)
    nt_stores = re.findall(r"global_store_dwordx4 .*\bnt\b", body)
    assert len(nt_stores) >= 8, len(nt_stores)
    assert "global_store_dwordx2" not in body and "global_store_dword " not in body
    # operand rows arrive as 16-byte loads; the grouped records are staged through LDS
    assert len(re.findall(r"global_load_dwordx4", body)) >= 32
    # straight-line FULL variant: the compiler counts outstanding memory operations instead of draining them all
    # before every store group -- at most a few full drains (loop boundaries), many exact counts
    waits = re.findall(r"s_waitcnt vmcnt\((\d+)\)", body)
    assert len([w for w in waits if int(w) > 0]) >= 8 and len([w for w in waits if int(w) == 0]) <= 6, waits
    res = _resources(usage, DOMINANT)
    assert res["scratch"] == 0 and res["vgpr_spill"] == 0 and res["vgprs"] <= 256 and res["occupancy"] >= 2, res


def test_gather_kernel_isa(asm):
    text, usage = asm
    body = _body(text, GATHER)
    assert "v_fma_f64" not in body
    assert len(re.findall(r"global_store_dwordx4 .*\bnt\b", body)) >= 1 and len(re.findall(r"global_load_dwordx4", body)) >= 4
    res = _resources(usage, GATHER)
    assert res["scratch"] == 0 and res["vgprs"] <= 64 and res["occupancy"] >= 8, res


def test_search_kernels_use_lds_and_cross_lane_ops(asm):
    text, _ = asm
    loc = [m for m in re.finditer(r"^(_ZN3ndi13locate_kernelId[^:]*):", text, flags=re.M)]
    assert loc
    staged = _body(text, "_ZN3ndi13locate_kernelIdLb1ELi4EEEvNS_10LocateArgsIT_EE")
    assert "ds_bpermute_b32" in staged            # top pyramid level bisected across lanes
    assert re.search(r"\bds_read", staged) and "flat_load" not in staged     # explicit LDS pointers, no flat loads
