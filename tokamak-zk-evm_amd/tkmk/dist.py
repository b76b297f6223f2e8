"""ctypes binding of libtkmk_dist.so (include/tkmk_dist.h): the sharded MSM / bivariate NTT over RCCL on device buffers, one
process per GPU.  The communicator id (128 bytes from rank 0) travels over whatever channel the host has; with torch.distributed
that is one broadcast_object_list (comm_from_torch).  The torch-based helpers of tkmk/sharding.py remain for the CPU (gloo)
rehearsal of the same partitioning."""
import ctypes
import os

import numpy as np

import tkmk

LIB_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "libtkmk_dist.so")
_lib = None

# every symbol include/tkmk_dist.h declares (tests/test_abi.py checks the library exports all of them)
SYMBOLS = ["tkmk_comm_unique_id", "tkmk_comm_init", "tkmk_comm_init_loopback", "tkmk_comm_is_loopback", "tkmk_comm_device_turn",
           "tkmk_comm_broadcast_host", "tkmk_comm_destroy", "tkmk_comm_rank",
           "tkmk_comm_size", "tkmk_dist_last_error", "tkmk_msm_sharded", "tkmk_msm_multi_ex_sharded", "tkmk_bintt_sharded",
           "tkmk_comm_all_gather_host", "tkmk_comm_all_gather_dev", "tkmk_comm_agree", "tkmk_comm_abort", "tkmk_dist_fwd_cols_to_rows",
           "tkmk_dist_inv_rows_to_cols", "tkmk_dist_rows_rotate", "tkmk_comm_ring_shift", "tkmk_comm_describe", "tkmk_dist_relayout_cols_to_rows",
           "tkmk_dist_relayout_rows_to_cols"]
SKIP_X_PASS, SKIP_Y_PASS = 1, 2          # TKMK_DIST_* of include/tkmk_dist.h


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("libtkmk_dist.so is not built (run __graft_entry__.build())")
        tkmk.lib()
        _lib = ctypes.CDLL(LIB_PATH)
        _lib.tkmk_dist_last_error.restype = ctypes.c_char_p
        _lib.tkmk_comm_destroy.argtypes = [ctypes.c_void_p]
        _lib.tkmk_comm_rank.argtypes = [ctypes.c_void_p]
        _lib.tkmk_comm_size.argtypes = [ctypes.c_void_p]
        _lib.tkmk_comm_is_loopback.argtypes = [ctypes.c_void_p]
        _lib.tkmk_comm_abort.argtypes = [ctypes.c_void_p]
        _lib.tkmk_comm_agree.argtypes = [ctypes.c_void_p, ctypes.c_int]
    return _lib


class DistError(RuntimeError):
    def __init__(self, code, where):
        self.code = code
        super().__init__("%s failed: %s (tkmk_error %d)" % (where, lib().tkmk_dist_last_error().decode(), code))


def _check(code, where):
    if code != 0:
        raise DistError(code, where)


def unique_id():
    buf = (ctypes.c_uint8 * 128)()
    _check(lib().tkmk_comm_unique_id(buf), "tkmk_comm_unique_id")
    return bytes(buf)


class Comm:
    def __init__(self, comm_id, world, rank, _handle=None):
        if _handle is None:
            h = ctypes.c_void_p()
            buf = (ctypes.c_uint8 * 128)(*comm_id)
            _check(lib().tkmk_comm_init(buf, int(world), int(rank), ctypes.byref(h)), "tkmk_comm_init")
        else:
            h = _handle
        self._h = h
        self.world, self.rank = world, rank

    @property
    def handle(self):
        return self._h

    def msm_multi_ex_sharded(self, jobs, bases_form=tkmk.BASES_CONVERTED, c=0, bitsize=0):
        """jobs as tkmk.msm_multi_ex, each describing THIS rank's share -> len(jobs) full results (144-byte canonical projective)"""
        cfg = tkmk.lib().tkmk_msm_default_config()
        cfg.are_scalars_on_device = cfg.are_points_on_device = True
        cfg.c, cfg.bitsize = c, bitsize
        arr = tkmk.msm_job_ex_array(jobs)
        out = np.empty(144 * len(jobs), np.uint8)
        _check(lib().tkmk_msm_multi_ex_sharded(self._h, arr, len(jobs), ctypes.byref(cfg), int(bases_form), tkmk._p(out)), "tkmk_msm_multi_ex_sharded")
        return out

    def msm_sharded(self, scalars, bases, n=None):
        """this rank's shard (host arrays or DeviceBuffers) -> the full result, 144-byte canonical projective, on every rank"""
        cfg = tkmk.lib().tkmk_msm_default_config()
        cfg.are_scalars_on_device, cfg.are_points_on_device = tkmk._on_dev(scalars), tkmk._on_dev(bases)
        n = tkmk._len(scalars) if n is None else n
        out = np.empty(144, np.uint8)
        _check(lib().tkmk_msm_sharded(self._h, tkmk._p(scalars), tkmk._p(bases), int(n), ctypes.byref(cfg), tkmk._p(out)), "tkmk_msm_sharded")
        return out

    def bintt_sharded(self, slab, x_size, y_size, inverse=False, coset_x=None, coset_y=None, out=None):
        """slab: DeviceBuffer with this rank's x-slab (overwritten) -> DeviceBuffer with this rank's y-slab"""
        out = tkmk.DeviceBuffer(32 * x_size * (y_size // self.world)) if out is None else out
        _check(lib().tkmk_bintt_sharded(self._h, tkmk._p(slab), ctypes.c_size_t(x_size), ctypes.c_size_t(y_size), 1 if inverse else 0,
                                        tkmk._p(coset_x), tkmk._p(coset_y), tkmk._p(out)), "tkmk_bintt_sharded")
        return out

    def all_gather_host(self, data):
        """data: bytes-like of one fixed length on every rank -> list of world byte strings"""
        send = np.frombuffer(bytes(data), np.uint8)
        recv = np.empty(send.size * self.world, np.uint8)
        _check(lib().tkmk_comm_all_gather_host(self._h, tkmk._p(send), ctypes.c_size_t(send.size), tkmk._p(recv)), "tkmk_comm_all_gather_host")
        return [recv[q * send.size:(q + 1) * send.size].tobytes() for q in range(self.world)]

    def describe(self):
        """this rank's line (dict): transport, size, rank, RCCL's own count / rank, RCCL version, device PCI bus id and UUID"""
        import json
        buf = ctypes.create_string_buffer(512)
        _check(lib().tkmk_comm_describe(self._h, buf, ctypes.c_size_t(512)), "tkmk_comm_describe")
        return json.loads(buf.value.decode())

    def describe_all(self):
        """every rank's line, gathered OVER THIS COMMUNICATOR (collective: all ranks call it)"""
        import json
        mine = json.dumps(self.describe()).encode().ljust(512, b" ")
        return [json.loads(b.decode()) for b in self.all_gather_host(mine)]

    def agree(self, status=0):
        _check(lib().tkmk_comm_agree(self._h, int(status)), "tkmk_comm_agree")

    def abort(self):
        lib().tkmk_comm_abort(self._h)

    def fwd_cols_to_rows(self, cols, in_x, in_y, x_size, y_size, flags=0):
        """cols: DeviceBuffer, this rank's in_x x (in_y / G) COLS matrix -> DeviceBuffer with its (x_size / G) x y_size ROWS slab"""
        out = tkmk.DeviceBuffer(32 * (x_size // self.world) * y_size)
        _check(lib().tkmk_dist_fwd_cols_to_rows(self._h, tkmk._p(cols), ctypes.c_size_t(in_x), ctypes.c_size_t(in_y), ctypes.c_size_t(x_size),
                                                ctypes.c_size_t(y_size), int(flags), tkmk._p(out)), "tkmk_dist_fwd_cols_to_rows")
        return out

    def inv_rows_to_cols(self, rows, x_size, y_size, flags=0):
        """rows: DeviceBuffer, this rank's ROWS slab (overwritten) -> DeviceBuffer with its x_size x (y_size / G) COLS matrix"""
        out = tkmk.DeviceBuffer(32 * x_size * (y_size // self.world))
        _check(lib().tkmk_dist_inv_rows_to_cols(self._h, tkmk._p(rows), ctypes.c_size_t(x_size), ctypes.c_size_t(y_size), int(flags), tkmk._p(out)),
               "tkmk_dist_inv_rows_to_cols")
        return out

    def relayout_cols_to_rows(self, cols, x_size, y_size, record_bytes):
        out = tkmk.DeviceBuffer(record_bytes * (x_size // self.world) * y_size)
        _check(lib().tkmk_dist_relayout_cols_to_rows(self._h, tkmk._p(cols), ctypes.c_size_t(x_size), ctypes.c_size_t(y_size), ctypes.c_size_t(record_bytes), tkmk._p(out)),
               "tkmk_dist_relayout_cols_to_rows")
        return out

    def relayout_rows_to_cols(self, rows, x_size, y_size, record_bytes):
        out = tkmk.DeviceBuffer(record_bytes * x_size * (y_size // self.world))
        _check(lib().tkmk_dist_relayout_rows_to_cols(self._h, tkmk._p(rows), ctypes.c_size_t(x_size), ctypes.c_size_t(y_size), ctypes.c_size_t(record_bytes), tkmk._p(out)),
               "tkmk_dist_relayout_rows_to_cols")
        return out

    def rows_rotate(self, slab, h, y_size, rot):
        out = tkmk.DeviceBuffer(32 * h * y_size)
        _check(lib().tkmk_dist_rows_rotate(self._h, tkmk._p(slab), ctypes.c_size_t(h), ctypes.c_size_t(y_size), ctypes.c_size_t(rot), tkmk._p(out)),
               "tkmk_dist_rows_rotate")
        return out

    def close(self):
        if self._h:
            lib().tkmk_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def loopback_comms(world):
    """world communicators over the LOOPBACK transport (virtual ranks in this process on this GPU; test transport): use each from its
    own thread (run_ranks)"""
    arr = (ctypes.c_void_p * world)()
    _check(lib().tkmk_comm_init_loopback(int(world), arr), "tkmk_comm_init_loopback")
    return [Comm(None, world, r, _handle=ctypes.c_void_p(arr[r])) for r in range(world)]


def run_ranks(comms, fn):
    """fn(comm) on one host thread per virtual rank (ctypes releases the GIL inside the entries) -> [fn's result per rank]; the first
    exception of any rank is re-raised after all threads have ended"""
    import threading
    res, err = [None] * len(comms), [None] * len(comms)

    def body(k):
        try:
            res[k] = fn(comms[k])
        except BaseException as e:      # noqa: BLE001
            err[k] = e
    threads = [threading.Thread(target=body, args=(k,)) for k in range(len(comms))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for e in err:
        if e is not None:
            raise e
    return res


def comm_from_torch(dist):
    """communicator over the ranks of an initialised torch.distributed group: rank 0's id goes out in one broadcast"""
    world, rank = dist.get_world_size(), dist.get_rank()
    box = [unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    return Comm(box[0], world, rank)
