"""ctypes mirror of the library's own partition mode (include/flame_hip.h flame_hip_comm_* / flame_hip_part_*;
csrc/part.cpp): one graph cut into world x parts_per_rank subdomains, halo records exchanged by RCCL inside the
library -- no torch.distributed involved.  `flame_ros_amd/dist.py` is the torch harness of the same scheme."""
import ctypes as C

import numpy as np

from . import lib as _l

ID_BYTES = 128
PEER_BLOB_BYTES = 128


def rccl_available():
    return bool(_l.load().flame_hip_rccl_available())


def unique_id():
    """Rank 0: the 128-byte id every rank passes to Communicator (hand it over by any means)."""
    buf = C.create_string_buffer(ID_BYTES)
    _l.check(_l.load().flame_hip_comm_get_unique_id(buf), "flame_hip_comm_get_unique_id")
    return buf.raw


class Communicator:
    """uid = None: a communicator WITHOUT RCCL (flame_hip_comm_create_local) -- its partitions exchange through the peer
    transport only (Partition.peer_blob / peer_connect between processes)."""

    def __init__(self, device, rank, world, uid):
        self._lib = _l.load()
        self._h = C.c_void_p()
        if uid is None:
            _l.check(self._lib.flame_hip_comm_create_local(C.byref(self._h), device, rank, world), "flame_hip_comm_create_local")
        else:
            self._uid = C.create_string_buffer(uid, ID_BYTES)
            _l.check(self._lib.flame_hip_comm_create(C.byref(self._h), device, rank, world, self._uid), "flame_hip_comm_create")
        self.rank, self.world = rank, world

    def info(self, key):
        """"rank", "world", "rccl_ranks" (ncclCommCount), "device", "shared_gpu"."""
        v = C.c_int64()
        _l.check(self._lib.flame_hip_comm_info(self._h, key.encode(), C.byref(v)), "flame_hip_comm_info(%s)" % key)
        return v.value

    def close(self):
        if self._h:
            self._lib.flame_hip_comm_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Partition:
    """comm = None: a host-only plan of rank `plan_rank` of `plan_world` (no device, no RCCL)."""

    def __init__(self, comm, pos, edges, alpha, beta, z, wgt, x0=None, parts_per_rank=1, halo_depth=8, plan_rank=0, plan_world=1):
        self._lib = _l.load()
        self._h = C.c_void_p()
        f = lambda a: None if a is None else np.ascontiguousarray(a, np.float32)  # noqa: E731
        pos = np.ascontiguousarray(pos, np.float32).reshape(-1, 2)
        edges = np.ascontiguousarray(edges, np.int32).reshape(-1, 2)
        self.V, self.E, self.k = len(pos), len(edges), parts_per_rank
        alpha, beta, z, wgt, x0 = f(alpha), f(beta), f(z), f(wgt), f(x0)
        _l.check(self._lib.flame_hip_part_create(C.byref(self._h), comm._h if comm else None, plan_rank, plan_world, parts_per_rank,
                                                 halo_depth, self.V, self.E, _p(pos), _p(edges), _p(alpha), _p(beta), _p(z),
                                                 _p(wgt), _p(x0)), "flame_hip_part_create")

    def step(self, params, n):
        _l.check(self._lib.flame_hip_part_solve(self._h, C.byref(params), n), "flame_hip_part_solve")

    def update_data(self, z, wgt, x0=None):
        """New frame on the unchanged topology (whole-graph arrays, caller's order); the state is reset."""
        f = lambda a: None if a is None else np.ascontiguousarray(a, np.float32)  # noqa: E731
        z, wgt, x0 = f(z), f(wgt), f(x0)
        _l.check(self._lib.flame_hip_part_update_data(self._h, _p(z), _p(wgt), _p(x0)), "flame_hip_part_update_data")

    def sync(self):
        _l.check(self._lib.flame_hip_part_sync(self._h), "flame_hip_part_sync")

    def costs(self, params):
        s, d = C.c_double(), C.c_double()
        _l.check(self._lib.flame_hip_part_costs(self._h, C.byref(params), C.byref(s), C.byref(d)), "flame_hip_part_costs")
        return s.value, d.value

    def gather_solution(self):
        x, w1, w2 = (np.empty(self.V, np.float32) for _ in range(3))
        q = np.empty((self.E, 3), np.float32)
        _l.check(self._lib.flame_hip_part_gather(self._h, _p(x), _p(w1), _p(w2), _p(q)), "flame_hip_part_gather")
        return x, w1, w2, q

    def set_option(self, key, value):
        """"transport" 0 RCCL / 1 peer (records written straight into the receivers' inboxes; collective over RCCL when the
        communicator has one), "time_exchanges", "pipeline"."""
        _l.check(self._lib.flame_hip_part_set_option(self._h, key.encode(), int(value)), "flame_hip_part_set_option(%s)" % key)

    def peer_blob(self):
        """This rank's inbox handle (128 bytes) for the peer transport between processes: gather every rank's, in rank order,
        and hand the concatenation to peer_connect()."""
        buf = C.create_string_buffer(PEER_BLOB_BYTES)
        _l.check(self._lib.flame_hip_part_peer_blob(self._h, buf), "flame_hip_part_peer_blob")
        return buf.raw

    def peer_connect(self, blobs):
        buf = C.create_string_buffer(bytes(blobs), len(blobs))
        _l.check(self._lib.flame_hip_part_peer_connect(self._h, buf), "flame_hip_part_peer_connect")

    def info(self, key, local_part=0):
        v = C.c_int64()
        _l.check(self._lib.flame_hip_part_info(self._h, key.encode(), local_part, C.byref(v)), "flame_hip_part_info(%s)" % key)
        return v.value

    def array(self, key, local_part=0):
        n = self._lib.flame_hip_part_array(self._h, key.encode(), local_part, None, 0)
        if n < 0:
            raise _l.FlameHipError(int(n), "flame_hip_part_array(%s)" % key)
        out = np.empty(n, np.int32)
        self._lib.flame_hip_part_array(self._h, key.encode(), local_part, _p(out), n)
        return out

    def close(self):
        if self._h:
            self._lib.flame_hip_part_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()
