"""Command-line driver with the argument surface of the reference's `preprocess` and `prove` binaries
(packages/backend/preprocess/src/main.rs:12-63, packages/backend/prove/src/main.rs:8-97):
--crs DIR --synthesizer-stat DIR --output DIR [--subcircuit-library DIR].
preprocess reads <lib>/setupParams.json, <synth>/permutation.json, <synth>/instance.json and writes <out>/preprocess.json;
prove additionally reads <lib>/subcircuitInfo.json, <lib>/r1cs/subcircuit{id}.r1cs, <synth>/placementVariables.json and
writes <out>/proof.json; both outputs are in the Solidity-verifier format.  One difference, stated rather than hidden: the CRS is read from <crs>/combined_sigma.tkcrs, the
flat TKCRS001 section payload the reference derives from its rkyv archive (tkmk/crs.py), not from sigma_preprocess.rkyv
itself.  Needs an MI355X: there is no CPU fallback.

usage: python -m tkmk.cli {preprocess|prove} --crs DIR --synthesizer-stat DIR --output DIR --subcircuit-library DIR
       python -m tkmk.cli setup --subcircuit-library DIR --output DIR [--fixed-tau]        (trusted-setup's flags; G1 side of the CRS)"""
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


FIXED_TAU = {   # Tau::gen_fixed (packages/backend/libs/src/field_structures/mod.rs:43-64); alpha == x there
    "x": 0x7234cd9b97845e0125e84ae3ae81354e004558d8c82a83425652bc7b9ed49f7d,
    "y": 0x6ed0eea55cbeeebdc7a41033ebd196ffecc1806fdbc13a8d41b8f1aa273a4037,
    "alpha": 0x7234cd9b97845e0125e84ae3ae81354e004558d8c82a83425652bc7b9ed49f7d,
    "gamma": 0x088dfe3d1b76775ec267d6d0e27b753ec904c76e0bc32ca8223dc2ae1a0ac6b4,
    "delta": 0x04b8ce26374c547d8722ac51f5ed1e0f9cb891c332c69c865d96af150189a818,
    "eta": 0x52eb2aeb35b72b94a19ea232e984850f2cda5542fdc10368955d8ac6274f8579,
}
FIXED_G1 = (0x0b001b4cc05fa01578be7d4e821d6ff58f2a05c584fba3cb31a37942dece65eadec9a878add2282f7c2513abb8d4ab05,   # setup/trusted-setup/src/main.rs:71-74
            0x15e237775397ed22eef43dd36cdca277c9cf6fa7e4ffff0a5bb4b20a82392caacf0f63fb6cdb02bccf2f5af14970d6b9)
FIXED_G2 = ("0x1116094a7c01d4fd8abcfea69c658c92c037765bee00556b8d4063c33540b316ac68a2d913d3adc3b43c7d7cc7505cfc17206c8ae661f247979b3f1daa7fb6d5f7ce9c17b5ed1d7e8b421a2508b3f09a603e6a5fab3fcde7364fd178d656ac36",   # main.rs:75-78
            "0x15bf297a4b9842fb1a3a6f2dbf6b94de06997b11b2f72436c22efbb48d2f74b0de7239ea182a2ee50c23ae3d0be6fdee09459611409874fe4b04b1a7e42cb84eb4ae01728dc55dbd1343fda8d0fe94a299fc757acc1d2602a49a005b4ff90190")
STD_G1 = (0x17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb,     # the standard generator
          0x08b3f481e3aaa0f1a09e30ed741d8ae4fcf5e095d5d00af600db18cb2c04b3edd03cc744a2888ae40caa232946c5e7e1)


def _setup(args):
    """trusted-setup's argument surface (setup/trusted-setup/src/main.rs:27-46): --subcircuit-library DIR --output DIR [--fixed-tau].
    Writes <out>/combined_sigma.tkcrs (tkmk/setup.py: G1 side on the device, the ten G2 points on the host)"""
    import secrets
    import time
    import numpy as np
    import tkmk
    from tkmk import g2, setup
    if tkmk.device_count() < 1:
        raise SystemExit("no HIP device: the MI355X backend has no CPU fallback")
    tkmk.set_device(0)
    aff = lambda xy: np.frombuffer(xy[0].to_bytes(48, "little") + xy[1].to_bytes(48, "little"), np.uint8).copy()   # noqa: E731
    if args.fixed_tau:
        print("Using hardcoded G1, G2 generators and tau")
        tau, g1, h2 = FIXED_TAU, aff(FIXED_G1), g2.from_hex_pair(*FIXED_G2)
    else:                                               # Tau::gen + a random G1 point = [h]G for a random h
        tau = {k: 1 + secrets.randbelow(setup.R - 1) for k in setup.TAU_FIELDS}
        h = tkmk.DeviceBuffer.from_host(setup._fr(1 + secrets.randbelow(setup.R - 1)))
        g1 = tkmk.g1_batch_scalar_mul_device(h, aff(STD_G1), 1).to_host()
        h2 = g2.scalar_mul(1 + secrets.randbelow(setup.R - 1), g2.STD_G2)
    t0 = time.perf_counter()
    _, path = setup.trusted_setup(args.subcircuit_library, args.output, tau, g1, h2)
    print("combined_sigma.tkcrs written to %s (%.3f s)" % (path, time.perf_counter() - t0))


def main(argv=None):
    ap = argparse.ArgumentParser(prog="tkmk.cli")
    sub = ap.add_subparsers(dest="cmd", required=True)
    for name in ("preprocess", "prove"):
        p = sub.add_parser(name)
        p.add_argument("--crs", required=True, metavar="PATH")
        p.add_argument("--synthesizer-stat", required=True, metavar="PATH")
        p.add_argument("--output", required=True, metavar="PATH")
        p.add_argument("--subcircuit-library", required=True, metavar="PATH")
    p = sub.add_parser("setup")
    p.add_argument("--output", required=True, metavar="PATH")
    p.add_argument("--subcircuit-library", required=True, metavar="PATH")
    p.add_argument("--fixed-tau", action="store_true")
    args = ap.parse_args(argv)
    {"preprocess": _preprocess, "prove": _prove, "setup": _setup}[args.cmd](args)


if __name__ == "__main__":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    main()
