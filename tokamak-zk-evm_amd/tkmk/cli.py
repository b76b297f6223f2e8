"""Command-line driver with the argument surface of the reference's `preprocess` and `prove` binaries
(packages/backend/preprocess/src/main.rs:12-63, packages/backend/prove/src/main.rs:8-97):
--crs DIR --synthesizer-stat DIR --output DIR [--subcircuit-library DIR].
preprocess reads <lib>/setupParams.json, <synth>/permutation.json, <synth>/instance.json and writes <out>/preprocess.json;
prove additionally reads <lib>/subcircuitInfo.json, <lib>/r1cs/subcircuit{id}.r1cs, <synth>/placementVariables.json and
writes <out>/proof.json; both outputs are in the Solidity-verifier format.  One difference, stated rather than hidden: the CRS is read from <crs>/combined_sigma.tkcrs, the
flat TKCRS001 section payload the reference derives from its rkyv archive (tkmk/crs.py), not from sigma_preprocess.rkyv
itself.  Needs an MI355X: there is no CPU fallback.

usage: python -m tkmk.cli {preprocess|prove} --crs DIR --synthesizer-stat DIR --output DIR --subcircuit-library DIR"""
import argparse
import json
import os
import sys


def _preprocess(args):
    import tkmk
    from tkmk import crs, proofio
    from tkmk.preprocess import Preprocess
    if tkmk.device_count() < 1:
        raise SystemExit("no HIP device: the MI355X backend has no CPU fallback")
    tkmk.set_device(0)                                              # check_device(): device id 0 (libs/src/utils/mod.rs:88-110)
    with open(os.path.join(args.subcircuit_library, "setupParams.json")) as f:
        sp = json.load(f)
    path = os.path.join(args.crs, "combined_sigma.tkcrs")
    if not os.path.exists(path):
        raise SystemExit("No reference string is found. Run the Setup first (expected %s)." % path)
    sections = crs.read_payload(path)
    sigma1, tables = crs.load_sigma1(sections, sp)
    with open(os.path.join(args.synthesizer_stat, "permutation.json")) as f:
        permutation = json.load(f)
    with open(os.path.join(args.synthesizer_stat, "instance.json")) as f:
        instance = json.load(f)
    pre = Preprocess.gen(sigma1, tables["gamma_inv_o_inst"], permutation, instance, sp)
    os.makedirs(args.output, exist_ok=True)
    proofio.write_json(os.path.join(args.output, "preprocess.json"), pre.convert_format_for_solidity_verifier())
    print("preprocess.json written to", args.output)


def _prove(args):
    import time
    import tkmk
    from tkmk import prove
    if tkmk.device_count() < 1:
        raise SystemExit("no HIP device: the MI355X backend has no CPU fallback")
    tkmk.set_device(0)
    t0 = time.perf_counter()
    print("Prover initialization...")
    try:
        _, _, _, _, times = prove.prove(args.subcircuit_library, args.synthesizer_stat, args.crs, args.output)
    except FileNotFoundError as e:
        raise SystemExit(str(e))
    for k in ("init.total", "prove0", "prove1", "prove2", "prove3", "prove4"):
        print("%-10s %.3f s" % (k, times[k]))
    dt = time.perf_counter() - t0
    print("Prove completed. Total elapsed time: %.3fs (%.0f ms)" % (dt, 1e3 * dt))


def main(argv=None):
    ap = argparse.ArgumentParser(prog="tkmk.cli")
    sub = ap.add_subparsers(dest="cmd", required=True)
    for name in ("preprocess", "prove"):
        p = sub.add_parser(name)
        p.add_argument("--crs", required=True, metavar="PATH")
        p.add_argument("--synthesizer-stat", required=True, metavar="PATH")
        p.add_argument("--output", required=True, metavar="PATH")
        p.add_argument("--subcircuit-library", required=True, metavar="PATH")
    args = ap.parse_args(argv)
    {"preprocess": _preprocess, "prove": _prove}[args.cmd](args)


if __name__ == "__main__":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    main()
