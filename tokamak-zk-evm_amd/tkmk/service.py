"""ctypes binding of libtkmk_prover.so (include/tkmk_prover.h): the resident prover — the native C++ host side
(host/tkmk_service.hpp) that keeps the circuit-static state (subcircuit library as device CSR, CRS in HBM) between proofs
and turns one synthesizer directory into one proof.json per call.  Work-alike of the `prove` binary's main
(packages/backend/prove/src/main.rs:27-97) without the per-process start-up.  bench.py's headline step is `Prover.prove`;
nothing in this path is Python beyond the call itself."""
import ctypes
import json
import os

import tkmk

LIB_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "libtkmk_prover.so")
# the -DTKMK_TESTING_MODE build of the same source: accepts fixed blinding scalars (the reference's `testing-mode` feature);
# the production library refuses them
TESTING_LIB_PATH = os.path.join(os.path.dirname(LIB_PATH), "libtkmk_prover_testing.so")
_libs = {}
TEST_PARTS, COEFFICIENT_BASIS = 1, 2        # TKMK_PROVE_* of include/tkmk_prover.h

# every symbol include/tkmk_prover.h declares (tests/test_abi.py checks the library exports all of them)
SYMBOLS = ["tkmk_prover_open", "tkmk_prover_open_sharded", "tkmk_prover_world_size", "tkmk_prover_prove", "tkmk_prover_prove_ex", "tkmk_prover_close", "tkmk_prover_free_string", "tkmk_prover_last_error",
           "tkmk_prover_crs_source"]


class ProveTiming(ctypes.Structure):
    _fields_ = [("parse_s", ctypes.c_double), ("upload_s", ctypes.c_double), ("build_s", ctypes.c_double), ("binding_s", ctypes.c_double),
                ("init_s", ctypes.c_double), ("prove_s", ctypes.c_double * 5), ("write_s", ctypes.c_double), ("total_s", ctypes.c_double)]

    def as_dict(self):
        d = {k: getattr(self, k) for k in ("parse_s", "upload_s", "build_s", "binding_s", "init_s", "write_s", "total_s")}
        d.update({"prove%d_s" % k: self.prove_s[k] for k in range(5)})
        d["rounds_s"] = sum(self.prove_s)
        return d


def lib(testing=False):
    path = TESTING_LIB_PATH if testing else LIB_PATH
    if path not in _libs:
        if not os.path.exists(path):
            raise ImportError(os.path.basename(path) + " is not built (run __graft_entry__.build()); there is no CPU fallback")
        tkmk.lib()                                  # libtkmk_hip.so first (the prover library links against it)
        l = ctypes.CDLL(path)
        l.tkmk_prover_last_error.restype = ctypes.c_char_p
        l.tkmk_prover_crs_source.restype = ctypes.c_char_p
        l.tkmk_prover_crs_source.argtypes = [ctypes.c_void_p]
        l.tkmk_prover_free_string.argtypes = [ctypes.c_void_p]
        l.tkmk_prover_close.argtypes = [ctypes.c_void_p]
        _libs[path] = l
    return _libs[path]


class ProverError(RuntimeError):
    def __init__(self, code, where, testing=False):
        self.code = code
        super().__init__("%s failed: %s (tkmk_error %d)" % (where, lib(testing).tkmk_prover_last_error().decode(), code))


class Prover:
    """with Prover(lib_dir, crs_dir) as p: proof, timing = p.prove(synth_dir, out_dir)
    testing=True binds libtkmk_prover_testing.so, the only build that takes `testing_mixer_json`."""

    def __init__(self, subcircuit_library_dir, crs_dir, testing=False, comm=None):
        """comm: a tkmk.dist.Comm — this process (or, over the loopback transport, this thread) is one rank of a sharded prover:
        tkmk_prover_open_sharded; every rank constructs and then proves with the same arguments"""
        h = ctypes.c_void_p()
        self._h = None
        self._lib = lib(testing)
        self.testing = testing
        self.comm = comm
        if comm is None:
            code = self._lib.tkmk_prover_open(os.fsencode(subcircuit_library_dir), os.fsencode(crs_dir), ctypes.byref(h))
        else:
            code = self._lib.tkmk_prover_open_sharded(comm.handle, os.fsencode(subcircuit_library_dir), os.fsencode(crs_dir), ctypes.byref(h))
        if code != 0:
            raise ProverError(code, "tkmk_prover_open" if comm is None else "tkmk_prover_open_sharded", testing)
        self._h = h

    @property
    def crs_source(self):
        return self._lib.tkmk_prover_crs_source(self._h).decode()

    def prove(self, synthesizer_dir, output_dir=None, testing_mixer_json=None, want_json=True, test_parts=False, coefficient_basis=False,
              want_boxes=False):
        """-> (proof.json document as a dict or None, timing dict[, commit boxes]).  testing_mixer_json: a file with fixed blinding
        scalars, for differential tests only.  test_parts / coefficient_basis / want_boxes: tkmk_prover_prove_ex's flags and its record
        of the (x_degree + 1) x (y_degree + 1) box of every commitment"""
        tm = ProveTiming()
        doc, boxes = ctypes.c_void_p(), ctypes.c_void_p()
        flags = (TEST_PARTS if test_parts else 0) | (COEFFICIENT_BASIS if coefficient_basis else 0)
        code = self._lib.tkmk_prover_prove_ex(self._h, os.fsencode(synthesizer_dir), None if output_dir is None else os.fsencode(output_dir),
                                              None if testing_mixer_json is None else os.fsencode(testing_mixer_json), ctypes.c_int(flags), ctypes.byref(tm),
                                              ctypes.byref(doc) if want_json else None, ctypes.byref(boxes) if want_boxes else None)
        if code != 0:
            raise ProverError(code, "tkmk_prover_prove_ex", self.testing)
        out = None
        if want_json:
            out = json.loads(ctypes.string_at(doc.value).decode())
            self._lib.tkmk_prover_free_string(doc)
        if want_boxes:
            b = json.loads(ctypes.string_at(boxes.value).decode())
            self._lib.tkmk_prover_free_string(boxes)
            return out, tm.as_dict(), b
        return out, tm.as_dict()

    def close(self):
        if self._h:
            self._lib.tkmk_prover_close(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
