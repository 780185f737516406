"""Short-row 1-D sweep (rows under 256 vectors): every formulation of the library on the same 4 GB output buffer.

For each (dtype, lanes) the two-kernel flat form (locate + eval_flat_kernel, round 3's only path for these shapes),
the fused query-order kernel (tables from L2 / from LDS, UNR x workgroup size) and the grouped short-row kernel are
timed end to end (wall clock over `reps` stream-ordered calls) and per stage (the library's HIP events), and every
variant's output is compared with the flat form's bit for bit.  The reference line is the long-row bucketed kernel on
the same buffer (lanes = 4096 / sizeof(T) * 8 ... i.e. 32 KiB rows), which is what VERDICT r3 asks to be within 0.7 of.

    python tools/short_rows_sweep.py [--quick] [--shapes ref] > profiles/r04_short_rows_sweep.jsonl
"""
import argparse
import json
import os
import sys
import time

os.environ["NDI_TUNE_LIVE"] = "1"
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g

pkg = g.load_package()
dev = torch.device("cuda:0")

KNOBS = ("NDI_SHORT_MODE", "NDI_FUSED_UNR", "NDI_FUSED_TB", "NDI_FUSED_LDS", "NDI_FUSED_WGS", "NDI_SHORT_CQ", "NDI_FUSED_PACK", "NDI_FUSED_DEBUG")


def set_knobs(**kw):
    for k in KNOBS:
        os.environ.pop(k, None)
    for k, v in kw.items():
        os.environ[k] = str(v)


def run(interp, qd, out, path, reps):
    interp.strategy.path = path
    interp.strategy.interp_array_into(interp, qd, out, async_launch=True)
    interp.strategy.finish()
    pkg.profile_enable(True)
    pkg.profile_read(reset=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        interp.strategy.interp_array_into(interp, qd, out, async_launch=True)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / reps
    interp.strategy.finish()
    prof = pkg.profile_read(reset=True)
    pkg.profile_enable(False)
    return wall, prof


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true", help="fewer variants")
    ap.add_argument("--bytes", type=float, default=4e9, help="output bytes per call")
    ap.add_argument("--n", type=int, default=1024)
    ap.add_argument("--lanes", type=str, default="8,32,64,128,256")
    ap.add_argument("--dtypes", type=str, default="float64,float32")
    ap.add_argument("--strategy", type=str, default="cubic")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--only", type=str, default="", help="run only the variants whose name starts with this")
    ap.add_argument("--debug-bits", type=str, default="", help="tuning build only (NDI_LIB=libndinterp_hip_tune.so): "
                    "comma-separated NDI_FUSED_DEBUG values, each variant is run once per value (bit 0 no search, "
                    "1 operands always from 8 hot intervals, 2 no stores, 3 no operand loads)")
    args = ap.parse_args()
    rng = np.random.default_rng(0)
    lanes = [int(v) for v in args.lanes.split(",")]
    for dname in args.dtypes.split(","):
        dt = np.dtype(dname)
        tdt = torch.float64 if dt == np.float64 else torch.float32
        el = dt.itemsize
        x = np.unique(rng.uniform(0, 1, 2 * args.n).astype(dt))[: args.n]
        xd = torch.as_tensor(x, device=dev)
        n = xd.numel()
        # reference line: the long-row bucketed kernel on the same buffer (32 KiB rows)
        Lref = 32768 // el
        for L in [Lref] + lanes:
            Q = int(min(args.bytes // (L * el), 2**31 - 1))
            yd = torch.rand((n, L), dtype=tdt, device=dev)
            strat = pkg.CubicSpline.new() if args.strategy == "cubic" else pkg.Linear.new()
            interp = pkg.Interp1DBuilder.new(yd).x(xd).strategy(strat).build()
            qd = (torch.rand(Q, dtype=tdt, device=dev) * (xd[-1] - xd[0]) * 0.999 + xd[0]).clamp(xd[0], xd[-1])
            out = torch.empty((Q, L), dtype=tdt, device=dev)
            ref = None
            variants = []
            if L == Lref:
                variants = [("bucketed_long", pkg.PATH_BUCKETED, {})]
            else:
                variants.append(("flat", pkg.PATH_GATHER, dict(NDI_SHORT_MODE=1)))
                unrs = (2,) if args.quick else (1, 2, 4)
                for unr in unrs:
                    variants.append((f"fused_l2_u{unr}", pkg.PATH_GATHER, dict(NDI_SHORT_MODE=2, NDI_FUSED_LDS=0, NDI_FUSED_UNR=unr, NDI_FUSED_PACK=0)))
                if L * el < 128:
                    for unr in unrs:
                        variants.append((f"fused_l2pack_u{unr}", pkg.PATH_GATHER, dict(NDI_SHORT_MODE=2, NDI_FUSED_LDS=0, NDI_FUSED_UNR=unr, NDI_FUSED_PACK=1)))
                if not args.quick and L * el < 128:
                    for tb in (512, 1024):
                        for unr in (2, 4):
                            variants.append((f"fused_l2pack_u{unr}_tb{tb}", pkg.PATH_GATHER,
                                             dict(NDI_SHORT_MODE=2, NDI_FUSED_LDS=0, NDI_FUSED_UNR=unr, NDI_FUSED_PACK=1, NDI_FUSED_TB=tb)))
                if not args.quick:
                    variants.append(("fused_l2_u2_tb512", pkg.PATH_GATHER, dict(NDI_SHORT_MODE=2, NDI_FUSED_LDS=0, NDI_FUSED_UNR=2, NDI_FUSED_TB=512)))
                    variants.append(("fused_l2_u2_wgs8", pkg.PATH_GATHER, dict(NDI_SHORT_MODE=2, NDI_FUSED_LDS=0, NDI_FUSED_UNR=2, NDI_FUSED_WGS=8)))
                tab = (3 * n - 2) * L * el if args.strategy == "cubic" else n * L * el
                if tab <= 150 * 1024:
                    for unr in unrs:
                        for tb in ((0,) if args.quick else (0, 256, 1024)):
                            variants.append((f"fused_lds_u{unr}_tb{tb}", pkg.PATH_GATHER,
                                             dict(NDI_SHORT_MODE=2, NDI_FUSED_LDS=1, NDI_FUSED_UNR=unr, NDI_FUSED_TB=tb)))
                if args.strategy == "cubic" and 2 * n * L * el <= 140 * 1024:   # {y, k} in LDS, a / b re-formed per item
                    for unr in unrs:
                        for tb in ((0,) if args.quick else (0, 256, 512, 1024)):
                            variants.append((f"fused_ldsyk_u{unr}_tb{tb}", pkg.PATH_GATHER,
                                             dict(NDI_SHORT_MODE=2, NDI_FUSED_LDS=2, NDI_FUSED_UNR=unr, NDI_FUSED_TB=tb)))
                for cq in ((64,) if args.quick else (16, 64)):
                    variants.append((f"grouped_cq{cq}", pkg.PATH_BUCKETED, dict(NDI_SHORT_MODE=3, NDI_SHORT_CQ=cq)))
                variants.append(("auto", pkg.PATH_AUTO, {}))
            if args.only:
                variants = [v for v in variants if v[0].startswith(args.only)]
            if args.debug_bits:
                variants = [(f"{nm}@dbg{b}", pth, dict(kn, NDI_FUSED_DEBUG=b)) for nm, pth, kn in variants
                            for b in args.debug_bits.split(",")]
            for name, path, knobs in variants:
                set_knobs(**knobs)
                out.fill_(-1.0)
                wall, prof = run(interp, qd, out, path, args.reps)
                rec = dict(dtype=dname, lanes=L, n=n, queries=Q, variant=name, ms=round(wall * 1e3, 4),
                           out_TBps=round(Q * L * el / wall / 1e12, 3), Gpts=round(Q * L / wall / 1e9, 1),
                           eval_ms=round(prof["eval_ms"] / max(args.reps, 1), 4),
                           locate_ms=round(prof["locate_ms"] / max(args.reps, 1), 4),
                           group_ms=round(prof["group_ms"] / max(args.reps, 1), 4), last_path=prof["last_path"])
                if L != Lref and not args.only:
                    if ref is None:
                        ref = out.clone()
                        rec["bit_exact_vs_flat"] = None
                    else:
                        rec["bit_exact_vs_flat"] = bool(torch.equal(out, ref))
                print(json.dumps(rec), flush=True)
            set_knobs()
            del out, ref, qd, interp, yd
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
