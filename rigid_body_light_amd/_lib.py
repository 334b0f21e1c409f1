"""ctypes view of librbl.so's device-pointer API (include/rbl.h section 3) for callers
that keep data resident on the GPU (bench.py, dist.py).  torch is used only as the
owner of device memory and streams; the pointers handed over are raw HIP pointers."""
import ctypes as C
import os

try:  # one HIP runtime per process: torch's bundled copy must be loaded first (see __init__.py)
    import torch as _torch  # noqa: F401
except ImportError:
    _torch = None

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        path = os.environ.get("RBL_LIBRARY") or os.path.join(_HERE, "librbl.so")   # override: A/B kernel builds
        if not os.path.exists(path):
            raise ImportError("librbl.so not built; run `python rigid_body_light_amd/build.py`")
        L = C.CDLL(path)
        vp, i64, dbl = C.c_void_p, C.c_int64, C.c_double
        L.rbl_create.restype = vp
        L.rbl_destroy.argtypes = [vp]
        L.rbl_last_error.restype = C.c_char_p
        L.rbl_last_error.argtypes = [vp]
        L.rbl_set_parameters.argtypes = [vp, dbl, dbl, dbl, dbl, vp, C.c_int]
        L.rbl_set_wall_pc.argtypes = [vp, C.c_int]
        L.rbl_set_config.argtypes = [vp, vp, vp, C.c_int]
        L.rbl_set_stream.argtypes = [vp, vp]
        L.rbl_apply_M_dev.argtypes = [vp, vp, vp, i64, i64, i64, vp]
        L.rbl_sync_bodies_dev.argtypes = [vp]
        L.rbl_prepare_dev.argtypes = [vp]
        L.rbl_positions_dev.argtypes = [vp, C.POINTER(C.c_void_p), C.POINTER(i64)]
        L.rbl_K_x_U_dev.argtypes = [vp, vp, vp]
        L.rbl_KT_x_Lam_dev.argtypes = [vp, vp, vp]
        L.rbl_apply_PC_dev.argtypes = [vp, vp, vp]
        L.rbl_apply_saddle_dev.argtypes = [vp, vp, vp]
        L.rbl_evolve_X_Q.argtypes = [vp, vp]
        L.rbl_get_config.argtypes = [vp, vp, vp]
        L.rbl_set_blk_pc.argtypes = [vp, C.c_int]
        L.rbl_apply_M_multi_dev.argtypes = [vp, vp, vp, i64, C.c_int, vp]
        L.rbl_apply_M_sym_dev.argtypes = [vp, vp, vp, i64, C.c_int, C.c_int, vp]
        L.rbl_apply_M_sym_multi_dev.argtypes = [vp, vp, vp, i64, C.c_int, C.c_int, C.c_int, vp]
        L.rbl_apply_M_sym_info.argtypes = [vp, i64, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(i64)]
        L.rbl_apply_M_sym_kernel.argtypes = [vp, i64, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_int]
        L.rbl_blob_positions_dev.argtypes = [vp, C.c_int, C.c_int, vp]
        L.rbl_rotne_prager_tensor_dev.argtypes = [vp, vp, i64, C.c_int, vp]
        L.rbl_cholesky_lower_dev.argtypes = [vp, vp, i64, C.c_int]
        L.rbl_trmv_lower_dev.argtypes = [vp, vp, i64, vp, vp]
        L.rbl_M_half_W_dev.argtypes = [vp, vp, i64, vp, C.c_int, vp]
        L.rbl_sync_check.argtypes = [vp]
        L.rbl_set_lanczos.argtypes = [vp, C.c_int, dbl]
        L.rbl_get_lanczos_report.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(dbl)]
        L.rbl_update_X_Q.argtypes = [vp, vp, vp, vp]
        L.rbl_step_deterministic.argtypes = [vp, vp, vp, C.c_int, dbl, C.c_int, C.POINTER(C.c_int), C.POINTER(dbl)]
        L.rbl_step_brownian.argtypes = [vp, vp, vp, vp, C.c_uint64, C.c_int, C.c_int, dbl, C.c_int, dbl, C.POINTER(C.c_int), C.POINTER(dbl)]
        L.rbl_block_solve_dev.argtypes = [vp, vp, vp, C.c_int]
        L.rbl_block_solve_range_dev.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.c_int]
        L.rbl_set_no_damp.argtypes = [vp, C.c_int]
        L.rbl_set_block_refresh.argtypes = [vp, C.c_int]
        L.rbl_gmres_saddle_dev.argtypes = [vp, vp, C.c_int, dbl, vp, C.c_int, C.POINTER(C.c_int), C.POINTER(dbl)]
        L.rbl_gmres_saddle_multi_dev.argtypes = [vp, vp, C.c_int, C.c_int, dbl, vp, C.POINTER(C.c_int), C.POINTER(dbl)]
        L.rbl_Kinv_x_V.argtypes = [vp, vp, vp]
        L.rbl_set_comm.argtypes = [vp, C.c_int, C.c_int, vp, vp]
        L.rbl_set_comm_ops.argtypes = [vp, C.c_int, C.c_int, vp, vp, vp]
        L.rbl_comm_unique_id.argtypes = [vp]
        L.rbl_comm_init_rccl.argtypes = [vp, vp, C.c_int, C.c_int]
        L.rbl_comm_finalize.argtypes = [vp]
        L.rbl_comm_info.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.rbl_comm_allreduce_dev.argtypes = [vp, vp, i64]
        L.rbl_comm_allgatherv_dev.argtypes = [vp, vp, C.POINTER(i64), C.POINTER(i64)]
        L.rbl_set_option.argtypes = [vp, C.c_int, i64]
        L.rbl_get_option.argtypes = [vp, C.c_int, C.POINTER(i64)]
        L.rbl_option_info.argtypes = [C.c_int, C.POINTER(C.c_char_p), C.POINTER(i64), C.POINTER(i64), C.POINTER(i64)]
        L.rbl_option_key.argtypes = [C.c_char_p]
        L.rbl_apply_saddle.argtypes = [vp, vp, vp]
        L.rbl_multi_body_pos_dev.argtypes = [vp, vp]
        L.rbl_RHS_and_Midpoint_dev.argtypes = [vp, vp, vp, vp, C.c_uint64, C.c_int, C.c_int, dbl, vp, vp, vp]
        L.rbl_set_timing.argtypes = [vp, C.c_int]
        L.rbl_reset_timings.argtypes = [vp]
        L.rbl_get_timings.argtypes = [vp, C.POINTER(dbl), C.POINTER(i64)]
        _LIB = L
    return _LIB


class RblError(RuntimeError):
    pass


class DeviceContext:
    """Owns one rbl_ctx bound to the current HIP device and a stream."""

    def __init__(self, a, eta, wall, cfg=None, dt=0.0, kBT=1.0, stream_ptr=None):
        import numpy as np
        self.L = lib()
        self.h = self.L.rbl_create()
        if not self.h:
            raise RblError("rbl_create failed")
        cfg = np.ascontiguousarray(np.zeros((1, 3)) if cfg is None else cfg, dtype=np.float64)
        self._chk(self.L.rbl_set_parameters(self.h, a, dt, kBT, eta, cfg.ctypes.data, cfg.shape[0]))
        self._chk(self.L.rbl_set_wall_pc(self.h, int(bool(wall))))
        self._chk(self.L.rbl_set_stream(self.h, stream_ptr))
        self._stream_ptr = int(stream_ptr or 0)

    def _chk(self, rc):
        if rc != 0:
            raise RblError("%s [rbl status %d]" % (self.L.rbl_last_error(self.h).decode(), rc))

    def set_config(self, X, Q):
        import numpy as np
        X = np.ascontiguousarray(X, dtype=np.float64).reshape(-1)
        Q = np.ascontiguousarray(Q, dtype=np.float64).reshape(-1)
        self._chk(self.L.rbl_set_config(self.h, X.ctypes.data, Q.ctypes.data, X.size // 3))

    def set_comm(self, sharded, native=None):
        """multi-GPU inside the library's solvers: `sharded` is a dist.ShardedMobility (rank, world, process group).  Every
        full mobility product of librbl's own GMRES / Lanczos / step drivers becomes this rank's share + one collective.
        native (default: whenever the process group moves device buffers, i.e. backend nccl): RCCL INSIDE librbl --
        rank 0's rbl_comm_unique_id is broadcast through the process group once, then rbl_comm_init_rccl; no Python runs
        between two products of a solve.  Otherwise (gloo rehearsals, several ranks sharing one GPU) the collectives are
        callbacks into torch.distributed, host-staged.  None switches it off.  A group of one rank keeps it on only when
        `sharded` was built with force_collectives (the world-1 RCCL test)."""
        import torch
        import torch.distributed as dist
        if sharded is None or not sharded.collectives:
            self._comm_cb = self._comm_cb2 = None
            self._chk(self.L.rbl_set_comm(self.h, 0, 1, None, None))
            return
        if native is None:
            native = not sharded.stage_cpu and sharded.device.type == "cuda"
        if native:
            ident = [None]
            if sharded.rank == 0:
                buf = C.create_string_buffer(128)
                if self.L.rbl_comm_unique_id(buf) != 0:
                    raise RblError("rbl_comm_unique_id failed (librccl not loadable?)")
                ident[0] = buf.raw
            dist.broadcast_object_list(ident, src=dist.get_global_rank(sharded.group, 0) if sharded.group is not None else 0,
                                       group=sharded.group)
            self._comm_cb = self._comm_cb2 = None
            self._chk(self.L.rbl_comm_init_rccl(self.h, ident[0], sharded.rank, sharded.world))
            return

        class _View:   # a raw device pointer as a torch tensor (no copy)
            def __init__(self, ptr, n):
                self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (ptr, False), "version": 2}

        views = {}     # (pointer, count) -> tensor view: librbl works on the same few buffers over and over

        def _view(ptr, count):
            t = views.get((ptr, count))
            if t is None:
                t = views[(ptr, count)] = torch.as_tensor(_View(ptr, int(count)), device=sharded.device)
            return t

        def _on_ctx_stream(fn):
            # rbl.h: a collective must be ordered on the CONTEXT's stream (the library enqueues producer and consumer
            # kernels there); torch issues collectives on its current stream, so make the context's stream current
            if sharded.device.type == "cuda" and torch.cuda.current_stream(sharded.device).cuda_stream != self._stream_ptr:
                with torch.cuda.stream(torch.cuda.ExternalStream(self._stream_ptr, device=sharded.device)):
                    fn()
            else:
                fn()

        def _allreduce(user, ptr, count):
            try:
                _on_ctx_stream(lambda: sharded.all_reduce_sum(_view(ptr, count)))
                return 0
            except Exception as e:      # never unwind through the C caller
                import sys
                print("rbl all-reduce callback failed: %r" % (e,), file=sys.stderr)
                return 1

        def _allgatherv(user, ptr, offs, cnts):
            try:
                W = sharded.world
                o = [int(offs[r]) for r in range(W)]
                n = [int(cnts[r]) for r in range(W)]
                _on_ctx_stream(lambda: sharded.all_gather_segments(lambda a, k: _view(ptr + 8 * a, k), o, n))
                return 0
            except Exception as e:
                import sys
                print("rbl all-gather callback failed: %r" % (e,), file=sys.stderr)
                return 1

        self._comm_cb = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64)(_allreduce)
        self._comm_cb2 = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64))(_allgatherv)
        self._chk(self.L.rbl_set_comm_ops(self.h, sharded.rank, sharded.world, C.cast(self._comm_cb, C.c_void_p),
                                          C.cast(self._comm_cb2, C.c_void_p), None))

    def comm_info(self):
        """(rank, world, kind) of the context's communicator; kind 0 none, 1 callbacks, 2 RCCL inside librbl"""
        r, w, k = C.c_int(0), C.c_int(1), C.c_int(0)
        self._chk(self.L.rbl_comm_info(self.h, C.byref(r), C.byref(w), C.byref(k)))
        return r.value, w.value, k.value

    def comm_finalize(self):
        self._chk(self.L.rbl_comm_finalize(self.h))
        self._comm_cb = self._comm_cb2 = None

    # -- named options (include/rbl.h RBL_OPT_*) -----------------------------------------
    def _opt_key(self, name):
        key = name if isinstance(name, int) else self.L.rbl_option_key(str(name).encode())
        if not key:
            raise RblError("unknown option %r" % (name,))
        return key

    def set_option(self, name, value):
        self._chk(self.L.rbl_set_option(self.h, self._opt_key(name), int(value)))

    def get_option(self, name):
        v = C.c_int64(0)
        self._chk(self.L.rbl_get_option(self.h, self._opt_key(name), C.byref(v)))
        return v.value

    def set_stream(self, stream_ptr):
        self._chk(self.L.rbl_set_stream(self.h, stream_ptr))
        self._stream_ptr = int(stream_ptr or 0)

    TIMING_PHASES = ("product", "per_body", "factor", "collective", "dense", "total")      # RBL_T_* of include/rbl.h

    def set_timing(self, on=True):
        """hipEvent brackets around the phases of librbl's own solvers (rbl_set_timing)"""
        self._chk(self.L.rbl_set_timing(self.h, int(bool(on))))

    def reset_timings(self):
        self._chk(self.L.rbl_reset_timings(self.h))

    def timings(self):
        """{phase: (milliseconds, brackets)} accumulated since the last reset (synchronises the stream)"""
        n = len(self.TIMING_PHASES)
        ms, calls = (C.c_double * n)(), (C.c_int64 * n)()
        self._chk(self.L.rbl_get_timings(self.h, ms, calls))
        return {k: (ms[i], calls[i]) for i, k in enumerate(self.TIMING_PHASES)}

    def apply_M(self, dF, dr, n_blobs, row_begin, row_end, dout):
        """dF, dr, dout: integer device addresses (tensor.data_ptr())."""
        self._chk(self.L.rbl_apply_M_dev(self.h, dF, dr, n_blobs, row_begin, row_end, dout))

    # -- device-resident rigid-body operators (own configuration) ----------------------
    def positions_ptr(self):
        p, n = C.c_void_p(), C.c_int64()
        self._chk(self.L.rbl_positions_dev(self.h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def prepare(self):
        """uploads, workspace growth and PC build now -> later operator calls are launch-only"""
        self._chk(self.L.rbl_prepare_dev(self.h))

    def K_x_U(self, dU, dout):
        self._chk(self.L.rbl_K_x_U_dev(self.h, dU, dout))

    def KT_x_Lam(self, dlam, dout):
        self._chk(self.L.rbl_KT_x_Lam_dev(self.h, dlam, dout))

    def apply_PC(self, din, dout):
        self._chk(self.L.rbl_apply_PC_dev(self.h, din, dout))

    def apply_saddle(self, dx, dout):
        self._chk(self.L.rbl_apply_saddle_dev(self.h, dx, dout))

    def evolve(self, U_host):
        import numpy as np
        U = np.ascontiguousarray(U_host, dtype=np.float64).reshape(-1)
        self._chk(self.L.rbl_evolve_X_Q(self.h, U.ctypes.data))

    def block_solve(self, din, dout, mode, body_begin=0, body_end=-1):
        """per-body Cholesky factors L L^T = M_body: mode 0 (L L^T)^-1, 1 L^-1, 2 L^-T, 3 L x; full-length blob
        vectors, only the bodies [body_begin, body_end) are factored, read and written (default: all)"""
        self._chk(self.L.rbl_block_solve_range_dev(self.h, din, dout, mode, int(body_begin), int(body_end)))

    def set_block_refresh(self, every):
        """keep the per-body Cholesky factors for `every` configuration changes (1 = rebuild after each)"""
        self._chk(self.L.rbl_set_block_refresh(self.h, int(every)))

    def set_no_damp(self, on):
        self._chk(self.L.rbl_set_no_damp(self.h, int(bool(on))))

    def step_deterministic(self, F_body, max_iter=20, rtol=None, slip=None, warm_start=False):
        """one deterministic time step inside librbl (solve + evolve) -> (iterations, residual estimate);
        warm_start 0..3: cold / previous solution / linear / quadratic extrapolation of the last solutions"""
        import numpy as np
        F = np.ascontiguousarray(F_body, dtype=np.float64).reshape(-1)
        sl = None if slip is None else np.ascontiguousarray(slip, dtype=np.float64).reshape(-1)
        it, res = C.c_int(0), C.c_double(0.0)
        self._chk(self.L.rbl_step_deterministic(self.h, F.ctypes.data, None if sl is None else sl.ctypes.data, int(max_iter),
                                                float(rtol or 0.0), int(warm_start), C.byref(it), C.byref(res)))
        return it.value, res.value

    def step_brownian(self, F_body, max_iter=20, rtol=None, slip=None, W=None, seed=0, method=2, split_rand=True,
                      delta=1.0e-4):
        """one stochastic midpoint step inside librbl -> (iterations, residual estimate)"""
        import numpy as np
        F = np.ascontiguousarray(F_body, dtype=np.float64).reshape(-1)
        sl = None if slip is None else np.ascontiguousarray(slip, dtype=np.float64).reshape(-1)
        Wh = None if W is None else np.ascontiguousarray(W, dtype=np.float64).reshape(-1)
        it, res = C.c_int(0), C.c_double(0.0)
        self._chk(self.L.rbl_step_brownian(self.h, F.ctypes.data, None if sl is None else sl.ctypes.data,
                                           None if Wh is None else Wh.ctypes.data, int(seed), int(method), int(bool(split_rand)),
                                           float(delta), int(max_iter), float(rtol or 0.0), C.byref(it), C.byref(res)))
        return it.value, res.value

    def gmres_saddle(self, d_rhs, max_iter, rtol, d_x, use_x0=False):
        """native right-preconditioned GMRES on the saddle operator -> (iterations, residual estimate);
        use_x0: d_x holds an initial guess (warm start from the previous time step)"""
        it, res = C.c_int(0), C.c_double(0.0)
        self._chk(self.L.rbl_gmres_saddle_dev(self.h, d_rhs, int(max_iter), float(rtol or 0.0), d_x, int(bool(use_x0)),
                                              C.byref(it), C.byref(res)))
        return it.value, res.value

    def gmres_saddle_multi(self, d_rhs, nrhs, max_iter, rtol, d_x):
        """nrhs right-hand sides in lock step (device vectors, one after the other) -> (iterations[nrhs], residuals[nrhs])"""
        it, res = (C.c_int * nrhs)(), (C.c_double * nrhs)()
        self._chk(self.L.rbl_gmres_saddle_multi_dev(self.h, d_rhs, int(nrhs), int(max_iter), float(rtol or 0.0), d_x, it, res))
        return list(it), list(res)

    def update_X_Q(self, U_host, n_bodies):
        """configuration displaced by U (displacement units), not committed -> (X, Q)"""
        import numpy as np
        U = np.ascontiguousarray(U_host, dtype=np.float64).reshape(-1)
        X = np.zeros(3 * n_bodies); Q = np.zeros(4 * n_bodies)
        self._chk(self.L.rbl_update_X_Q(self.h, U.ctypes.data, X.ctypes.data, Q.ctypes.data))
        return X.reshape(-1, 3), Q.reshape(-1, 4)

    def Kinv_x_V(self, V_host, n_bodies):
        """K^-1 V (reference :406): V[3 N_blobs] host -> [6 N_bodies] host (O(N) host work)"""
        import numpy as np
        V = np.ascontiguousarray(V_host, dtype=np.float64).reshape(-1)
        out = np.zeros(6 * n_bodies)
        self._chk(self.L.rbl_Kinv_x_V(self.h, V.ctypes.data, out.ctypes.data))
        return out

    def RHS_and_Midpoint(self, dSlip, dForce, dW, seed, method, split_rand, delta, dRHS, n_bodies):
        """device-vector form of the reference's RHS_and_Midpoint -> (X_half, Q_half) on the host"""
        import numpy as np
        X = np.zeros(3 * n_bodies); Q = np.zeros(4 * n_bodies)
        self._chk(self.L.rbl_RHS_and_Midpoint_dev(self.h, dSlip, dForce, dW, seed, method, int(split_rand), delta,
                                                  dRHS, X.ctypes.data, Q.ctypes.data))
        return X.reshape(-1, 3), Q.reshape(-1, 4)

    def get_config(self, n_bodies):
        import numpy as np
        X = np.zeros(3 * n_bodies); Q = np.zeros(4 * n_bodies)
        self._chk(self.L.rbl_get_config(self.h, X.ctypes.data, Q.ctypes.data))
        return X.reshape(-1, 3), Q.reshape(-1, 4)

    def apply_M_multi(self, dF, dr, n_blobs, nrhs, dout):
        """nrhs vectors, column-major (3 n_blobs) x nrhs; >= 4 go through the fp64-MFMA kernel."""
        self._chk(self.L.rbl_apply_M_multi_dev(self.h, dF, dr, n_blobs, nrhs, dout))

    def apply_M_sym(self, dF, dr, n_blobs, i_first, i_step, dout):
        """partial product over share i_first of i_step of the tile rows (sum over the shares = full U)."""
        self._chk(self.L.rbl_apply_M_sym_dev(self.h, dF, dr, n_blobs, i_first, i_step, dout))

    def apply_M_sym_multi(self, dF, dr, n_blobs, nrhs, i_first, i_step, dout):
        """the same for nrhs = 1 or 2 vectors ([nrhs][3 n_blobs]); two vectors share the pair coefficients"""
        self._chk(self.L.rbl_apply_M_sym_multi_dev(self.h, dF, dr, n_blobs, nrhs, i_first, i_step, dout))

    def apply_M_sym_info(self, n_blobs, i_step=1, nrhs=1):
        """(rows per lane NI, column tiles per work unit, slab workspace bytes) of the symmetric kernel launch"""
        ni, ch, wb = C.c_int(0), C.c_int(0), C.c_int64(0)
        self._chk(self.L.rbl_apply_M_sym_info(self.h, n_blobs, i_step, nrhs, C.byref(ni), C.byref(ch), C.byref(wb)))
        return ni.value, ch.value, wb.value

    def apply_M_sym_kernel(self, n_blobs, wall, i_step=1, nrhs=1):
        """name of the kernel instantiation a symmetric product of that size launches under this context's options"""
        buf = C.create_string_buffer(64)
        self._chk(self.L.rbl_apply_M_sym_kernel(self.h, n_blobs, i_step, nrhs, 1 if wall else 0, buf, 64))
        return buf.value.decode()

    def blob_positions(self, body_begin, body_end, dout):
        self._chk(self.L.rbl_blob_positions_dev(self.h, body_begin, body_end, dout))

    def multi_body_pos(self, dout):
        """all blob positions into a device vector; on a row-split multi-GPU context: own bodies + one all-gather"""
        self._chk(self.L.rbl_multi_body_pos_dev(self.h, dout))

    def build_M(self, dr, n_blobs, scale_damp, dout):
        self._chk(self.L.rbl_rotne_prager_tensor_dev(self.h, dr, n_blobs, int(scale_damp), dout))

    def cholesky(self, dM, n, zero_upper=False):
        self._chk(self.L.rbl_cholesky_lower_dev(self.h, dM, n, int(zero_upper)))

    def trmv_lower(self, dL, n, dW, dout):
        self._chk(self.L.rbl_trmv_lower_dev(self.h, dL, n, dW, dout))

    def M_half_W(self, dr, n_blobs, dW, method, dout):
        self._chk(self.L.rbl_M_half_W_dev(self.h, dr, n_blobs, dW, {"cholesky": 0, "lanczos": 1, "lanczos_pc": 2}[method], dout))

    def set_lanczos(self, max_iter, tol):
        self._chk(self.L.rbl_set_lanczos(self.h, max_iter, tol))

    def lanczos_report(self):
        it, res = C.c_int(0), C.c_double(0.0)
        self.L.rbl_get_lanczos_report(self.h, C.byref(it), C.byref(res))
        return it.value, res.value

    def sync_check(self):
        self._chk(self.L.rbl_sync_check(self.h))

    def close(self):
        if self.h:
            self.L.rbl_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
