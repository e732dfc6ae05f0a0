"""ctypes binding of the C ABI in include/segx.h (libsegx.so, hand-written HIP for gfx950).

`lib()` loads the in-tree HIP build `segtran_amd/lib/libsegx.so` and raises if it is missing: the
product path has NO fallback (no oracle, no eager-PyTorch replacement).  The class itself is
path-parameterised only so that tests can point it at the fiber-emulated build of the SAME sources
(tests/hipemu) and exercise kernel index math on CPU tensors.
"""
import ctypes
import os
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'lib', 'libsegx.so')

c_f = ctypes.c_float
c_i = ctypes.c_int32
c_l = ctypes.c_int64
c_u = ctypes.c_uint64
c_p = ctypes.c_void_p


class GemmDesc(ctypes.Structure):
    _fields_ = [('M', c_i), ('N', c_i), ('K', c_i), ('nb0', c_i), ('nb1', c_i),
                ('a_b0', c_l), ('a_b1', c_l), ('a_m', c_l), ('a_k', c_l),
                ('b_b0', c_l), ('b_b1', c_l), ('b_n', c_l), ('b_k', c_l),
                ('c_b0', c_l), ('c_b1', c_l), ('c_m', c_l),
                ('alpha', c_f), ('epilogue', c_i), ('bias_mode', c_i), ('bias_b1', c_l),
                ('bias', c_p), ('aux', c_p), ('gmax', c_p),
                ('dropout_p', c_f), ('seed', c_u), ('offset', c_u),
                ('splitk', c_i), ('workspace', c_p), ('tile', c_i), ('bias_b0', c_l), ('batch_reduce', c_i), ('engine', c_i),
                ('b_planes', c_p), ('bp_b0', c_l), ('bp_b1', c_l), ('resid', c_p)]


EPI_NONE, EPI_GELU = 0, 1
(TILE_AUTO, TILE_128x128, TILE_64x64, TILE_128x32, TILE_32x128, TILE_64x128, TILE_256x128, TILE_WS128x128, TILE_WS128x256, TILE_WS64x256,
 TILE_WS96x256, TILE_WS256x96, TILE_SKINNY_NT) = range(13)
BIAS_NONE, BIAS_N, BIAS_M = 0, 1, 2


def _ptr(t):
    return None if t is None else t.data_ptr()       # a plain int: ctypes converts it for a c_void_p parameter without an object per argument


# the current stream's handle without building a torch.cuda.Stream object per launch (a private but long-standing entry point; the public path is the fallback)
_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)


class SegxLib:
    def __init__(self, path=LIB_PATH):
        if not os.path.exists(path):
            raise RuntimeError('libsegx not built: %s is missing '
                               '(run `python -c "import __graft_entry__ as g; g.build()"`)' % path)
        self.path = path
        self.c = ctypes.CDLL(path)
        self.c.segx_last_error.argtypes = [ctypes.c_char_p, c_i]
        self.c.segx_version.restype = c_i
        for name in ('segx_gemm_f32', 'segx_gemm_plan', 'segx_gemm_plan_model'):     # (A, B, C | desc, desc | tile*, stream | splitk*): pointers all
            fn = getattr(self.c, name)
            fn.argtypes = [c_p] * 5
            fn.restype = c_i
        self.emulated = 'emu' in os.path.basename(path)
        self.force_tile = None           # tools/gemm_bench.py: override the library's tile choice
        self.gemm_prof = None            # bench.py: list of (start_event, end_event, flops) per GEMM launch
        kinds = {'p': c_p, 'i': c_i, 'l': c_l, 'f': c_f, 'u': c_u}
        for name, sig in _SIGS.items():
            fn = getattr(self.c, name)
            fn.argtypes = [kinds[k] for k in sig]
            fn.restype = c_l if name.endswith(('_floats', '_rows', '_splitk', '_elems', '_chunks', '_nparts')) else c_i

    # ---- tile engine -------------------------------------------------------------------------
    ENGINES = {'f32': 0, 'x6': 1}

    def set_engine(self, name):
        """'f32': every GEMM / implicit-GEMM convolution on v_mfma_f32_32x32x2_f32 (a k-ordered fp32 fmaf chain); 'x6': eligible ones
        (float4-legal operands, > 48 rows on both sides) on the bf16 matrix core with the 3-way operand split (fp32-equivalent).
        Returns the previous engine name."""
        prev = self.c.segx_tune(4, self.ENGINES[name])
        if prev < 0:
            raise RuntimeError('segx_tune(4): engine %r rejected' % name)
        return 'x6' if prev == 1 else 'f32'

    def set_rng_base(self, t):
        """t: int64 [1] device tensor (kept alive by the caller) holding the device-side base of every dropout stream, or None."""
        self.check(self.c.segx_set_rng_base(_ptr(t)), 'segx_set_rng_base')

    def rng_advance(self, t, span):
        self.check(self.c.segx_rng_advance(_ptr(t), int(span), self.stream(t)), 'segx_rng_advance')

    def x6_launches(self):
        """launches that ran on the bf16x6 engine since the last call"""
        return int(self.c.segx_tune(5, 0))

    # ---- team exchange: loud failure (include/segx.h: segx_team_status) ---------------------------------
    def team_check(self):
        """Raise if a team exchange of an earlier launch timed out (the team BatchNorm kernels then handed NaN statistics to their members).  Reads
        one word of pinned host memory: no device synchronisation; called by TrainStep / BertAdam.step on every step."""
        n = int(self.c.segx_team_status(1))
        if n:
            raise RuntimeError('libsegx: %d team exchange(s) of the team BatchNorm kernels timed out -- their workgroups were not co-resident (another kernel '
                               'holding the compute units, a CU mask, a partitioned device?).  The affected launches produced NaN statistics.  '
                               'segx_tune(3, 1) selects the two-launch BatchNorm, which needs no exchange.' % n)

    def team_cap(self):
        return int(self.c.segx_team_cap())

    # ---- plumbing -----------------------------------------------------------------------------
    def stream(self, t):
        if t.is_cuda:
            if _raw_stream is not None:
                idx = t.device.index
                return _raw_stream(idx if idx is not None else torch.cuda.current_device())
            return torch.cuda.current_stream(t.device).cuda_stream
        return 0

    def check(self, rc, what):
        if rc != 0:
            buf = ctypes.create_string_buffer(512)
            self.c.segx_last_error(buf, 512)
            raise RuntimeError('%s failed (rc=%d): %s' % (what, rc, buf.value.decode()))

    def _chk_t(self, *ts):
        dev = None
        for t in ts:
            if t is None:
                continue
            if not self.emulated and not t.is_cuda:
                raise RuntimeError('libsegx (HIP) called with a CPU tensor')
            dev = dev or t.device
            assert t.device == dev, 'tensors on different devices'

    # ---- GEMM -----------------------------------------------------------------------------------
    def gemm(self, A, B, C, M, N, K, a_strides, b_strides, c_strides, nb=(1, 1), alpha=1.0, bias=None,
             bias_mode=BIAS_NONE, bias_b1=0, bias_b0=0, epilogue=EPI_NONE, aux=None, gmax=None, dropout_p=0.0, seed=0,
             offset=0, splitk=1, workspace=None, tile=TILE_AUTO, batch_reduce=False, engine=None, b_planes=None, resid=None):
        """C[z][m][n] = epi(alpha * sum_k A[z][m][k] B[z][n][k] + bias).  Strides in elements:
        a_strides = (b0, b1, m, k); b_strides = (b0, b1, n, k); c_strides = (b0, b1, m).
        splitk = 0: take tile and split factor from segx_gemm_plan and allocate the slab workspace here.
        batch_reduce: C is ONE [M, N] matrix = the sum over all batch members (see segx_gemm_desc.batch_reduce).
        engine: None = the process default (set_engine), 'f32' / 'x6' = the tile engine of THIS call (segx_gemm_desc.engine).
        b_planes: (planes tensor from x6_presplit(B ...), bp_b0, bp_b1) -- B split ahead of time (segx_gemm_desc.b_planes).  Bit-identical results;
        measured (tools/pre_bench.py, profiles/r03_ad_pre_bench.txt): +-0 on the 1792-wide projections, +5..9 % on 896-wide ones on the 256 x 128
        kernel (which the planner no longer picks for them), so the model code does not use it."""
        self._chk_t(A, B, C, bias, aux, gmax, workspace, resid)
        d = GemmDesc()
        d.resid = _ptr(resid)
        d.M, d.N, d.K, d.nb0, d.nb1 = M, N, K, nb[0], nb[1]
        d.a_b0, d.a_b1, d.a_m, d.a_k = a_strides
        d.b_b0, d.b_b1, d.b_n, d.b_k = b_strides
        d.c_b0, d.c_b1, d.c_m = c_strides
        d.alpha, d.epilogue, d.bias_mode, d.bias_b1, d.bias_b0 = alpha, epilogue, bias_mode, bias_b1, bias_b0
        d.bias, d.aux, d.gmax = _ptr(bias), _ptr(aux), _ptr(gmax)
        d.dropout_p, d.seed, d.offset = dropout_p, seed, offset
        if b_planes is not None:
            self._chk_t(b_planes[0])
            d.b_planes, d.bp_b0, d.bp_b1 = _ptr(b_planes[0]), b_planes[1], b_planes[2]
        d.engine = 0 if engine is None else 1 + self.ENGINES[engine]        # before planning: the plan is made for the engine of THIS call
        d.batch_reduce = 1 if batch_reduce else 0
        if splitk == 0:
            t, sk = c_i(0), c_i(0)
            self.check(self.c.segx_gemm_plan(_ptr(A), _ptr(B), ctypes.byref(d), ctypes.byref(t), ctypes.byref(sk)), 'segx_gemm_plan')
            tile, splitk = t.value, sk.value
            workspace = torch.empty(splitk * nb[0] * nb[1] * M * N, dtype=torch.float32, device=C.device) if (splitk > 1 or batch_reduce) else None
        d.splitk, d.workspace = splitk, _ptr(workspace)
        d.tile = self.force_tile if self.force_tile is not None else tile
        if self.gemm_prof is not None and C.is_cuda:
            # HIP events on the launch stream (torch's current stream IS the stream handed to the kernel); the last field records which
            # tile engine the library chose for this launch (knob 5 counts bf16x6 launches)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            self.x6_launches()
            e0.record()
            rc = self.c.segx_gemm_f32(_ptr(A), _ptr(B), _ptr(C), ctypes.byref(d), self.stream(C))
            e1.record()
            self.gemm_prof.append((e0, e1, 2.0 * M * N * K * nb[0] * nb[1], (M, N, K, nb[0] * nb[1], a_strides[3] == 1, b_strides[3] == 1, splitk, d.tile),
                                   self.x6_launches() > 0))
        else:
            rc = self.c.segx_gemm_f32(_ptr(A), _ptr(B), _ptr(C), ctypes.byref(d), self.stream(C))
        self.check(rc, 'segx_gemm_f32')

    def x6_presplit(self, W, rows, K, s_row, s_k, nb=(1, 1), s_b=(0, 0)):
        """Three-plane bf16 image of an fp32 operand (segx_x6_presplit) for gemm(b_planes=...): returns (planes, bp_b0, bp_b1).  A batch dimension
        with stride 0 (operand shared over it) is split once."""
        e0, e1 = (nb[0] if s_b[0] else 1), (nb[1] if s_b[1] else 1)
        planes = torch.empty(self.c.segx_x6_presplit_elems(rows, K, e0, e1), dtype=torch.int16, device=W.device)
        self._chk_t(W)
        self.check(self.c.segx_x6_presplit(_ptr(W), rows, K, s_row, s_k, e0, e1, s_b[0], s_b[1], _ptr(planes), self.stream(W)), 'segx_x6_presplit')
        per = 3 * rows * K
        return planes, (e1 * per if s_b[0] else 0), (per if s_b[1] else 0)

    # ---- token row kernels (tokens.hip) ---------------------------------------------------------
    def _call(self, name, ref, *args):
        """args: tensors (-> device pointers), None (-> NULL) or python scalars; stream appended."""
        conv = []
        dev = None
        hip = not self.emulated
        for a in args:                                       # one pass: the checks of _chk_t + contiguity + pointer extraction
            if isinstance(a, torch.Tensor):
                if hip and not a.is_cuda:
                    raise RuntimeError('libsegx (HIP) called with a CPU tensor')
                if dev is None:
                    dev = a.device
                elif a.device != dev:
                    raise AssertionError('tensors on different devices')
                assert a.is_contiguous(), name + ': non-contiguous tensor'
                conv.append(a.data_ptr())
            else:
                conv.append(a)
        rc = getattr(self.c, name)(*conv, self.stream(ref))
        if rc:
            self.check(rc, name)

    def colreduce_ws(self, rows, C, nout):
        return int(self.c.segx_colreduce_ws_floats(rows, C, nout))

    def softmax_fwd(self, S, P, Pdrop, rows, L, clip, gmax, p, seed, offset):
        self._call('segx_softmax_fwd', S, S, P, Pdrop, rows, L, clip, gmax, p, seed, offset)

    def softmax_bwd(self, P, dPd, S, dS, rows, L, clip, gmax, p, seed, offset):
        self._call('segx_softmax_bwd', P, P, dPd, S, dS, rows, L, clip, gmax, p, seed, offset)

    def posbias_fwd(self, S, out, table, nmat, N, geom, weight, clip, gmax):
        self._chk_t(S, out, table, gmax)
        rc = self.c.segx_posbias_fwd(_ptr(S), _ptr(out), _ptr(table), nmat, N, self._geom(geom), weight, clip, _ptr(gmax), self.stream(out))
        self.check(rc, 'segx_posbias_fwd')

    def posbias_bwd(self, dOut, S, dS, dtable, nmat, N, geom, weight, clip):
        self._chk_t(dOut, S, dS, dtable)
        rc = self.c.segx_posbias_bwd(_ptr(dOut), _ptr(S), _ptr(dS), _ptr(dtable), nmat, N, self._geom(geom), weight, clip, self.stream(dtable))
        self.check(rc, 'segx_posbias_bwd')

    def layernorm_fwd(self, X, w, b, Y, mean, rstd, rows, C, eps):
        self._call('segx_layernorm_fwd', X, X, w, b, Y, mean, rstd, rows, C, eps)

    def layernorm_bwd(self, dY, X, w, mean, rstd, dX, rows, C):
        self._call('segx_layernorm_bwd', X, dY, X, w, mean, rstd, dX, rows, C)

    def colsum(self, X, out, ws, rows, C):
        self._call('segx_colsum', X, X, out, ws, rows, C)

    def ln_param_grad(self, dY, X, mean, rstd, dw, db, ws, rows, C):
        self._call('segx_ln_param_grad', X, dY, X, mean, rstd, dw, db, ws, rows, C)

    def rowsum(self, X, out, R, S):
        self._call('segx_rowsum', X, X, out, R, S)

    def sum(self, x, n, out, ws, scale=1.0):
        self._call('segx_sum', x, x, n, out, ws, scale)

    def prenorm_fwd(self, X, w1, b1, pos, pos_ld, pw, mask, Y, stats, B, N, C, eps, p, seed, offset):
        self._call('segx_prenorm_fwd', X, X, w1, b1, pos, pos_ld, pw, mask, Y, stats, B, N, C, eps, p, seed, offset)

    def prenorm_bwd(self, dY, X, w1, b1, pos, pos_ld, pw, mask, stats, dX, dU, B, N, C, p, seed, offset):
        self._call('segx_prenorm_bwd', X, dY, X, w1, b1, pos, pos_ld, pw, mask, stats, dX, dU, B, N, C, p, seed, offset)

    def prenorm_bwd_all_ws(self, N, C):
        return int(self.c.segx_prenorm_bwd_all_ws_floats(N, C))

    def prenorm_bwd_all(self, dY, X, w1, b1, pos, pos_ld, pos_weight, mask, stats, dX, dsum, dw, db, ws, B, N, C, p, seed, offset):
        self._call('segx_prenorm_bwd_all', X, dY, X, w1, b1, pos, pos_ld, pos_weight, mask, stats, dX, dsum, dw, db, ws, B, N, C, p, seed, offset)

    def posembed_fwd(self, posn, Wp, bp, out, stats, N, C, pd, eps):
        self._call('segx_posembed_fwd', out, posn, Wp, bp, out, stats, N, C, pd, eps)

    def posembed_bwd(self, dOut, posn, Wp, bp, stats, dZ, N, C, pd):
        self._call('segx_posembed_bwd', dOut, dOut, posn, Wp, bp, stats, dZ, N, C, pd)

    def modes_aggr_fwd(self, Z, lnw, lnb, wa, ba, Y, stats, Mo, R, F, eps, p, seed, offset):
        self._call('segx_modes_aggr_fwd', Z, Z, lnw, lnb, wa, ba, Y, stats, Mo, R, F, eps, p, seed, offset)

    def modes_aggr_bwd(self, dY, Z, lnw, lnb, wa, stats, dZ, dscore, Mo, R, F, p, seed, offset):
        self._call('segx_modes_aggr_bwd', Z, dY, Z, lnw, lnb, wa, stats, dZ, dscore, Mo, R, F, p, seed, offset)

    def modes_aggr_param_grad(self, dY, Z, lnw, lnb, wa, stats, dscore, dlnw, dlnb, dwa, ws, Mo, R, F, p, seed, offset):
        self._call('segx_modes_aggr_param_grad', Z, dY, Z, lnw, lnb, wa, stats, dscore, dlnw, dlnb, dwa, ws, Mo, R, F,
                   p, seed, offset)

    def modes_aggr_bwd_all_ws(self, Mo, R, F):
        return int(self.c.segx_modes_aggr_bwd_all_ws_floats(Mo, R, F))

    def modes_aggr_bwd_all(self, dY, Z, lnw, lnb, wa, stats, dZ, dscore, dlnw, dlnb, dwa, ws, Mo, R, F, p, seed, offset):
        self._call('segx_modes_aggr_bwd_all', Z, dY, Z, lnw, lnb, wa, stats, dZ, dscore, dlnw, dlnb, dwa, ws, Mo, R, F, p, seed, offset)

    def gelu_bwd_colsum_ws(self, rows, N):
        return int(self.c.segx_gelu_bwd_colsum_ws_floats(rows, N))

    def gelu_bwd_colsum(self, dH, T, dT, colsum, ws, rows, N, p, seed, offset):
        self._call('segx_gelu_bwd_colsum', dH, dH, T, dT, colsum, ws, rows, N, p, seed, offset)

    def gelu_bwd(self, dH, T, dT, n, p, seed, offset):
        self._call('segx_gelu_bwd', T, dH, T, dT, n, p, seed, offset)

    # ---- train-step glue (train.hip) --------------------------------------------------------------
    def loss_ws(self, B, C):
        return int(self.c.segx_loss_ws_floats(B, C))

    def seg_loss_fwd(self, logits, mask, pw, cw, out, ws, B, C, S, dice_w):
        self._call('segx_seg_loss_fwd', logits, logits, mask, pw, cw, out, ws, B, C, S, dice_w)

    def seg_loss_bwd(self, logits, mask, pw, cw, ws, gout, dlogits, B, C, S, dice_w):
        self._call('segx_seg_loss_bwd', logits, logits, mask, pw, cw, ws, gout, dlogits, B, C, S, dice_w)

    # ---- backbone kernels (backbone.hip) ------------------------------------------------------------
    def bn_ws(self, B, C, S=0):
        return int(self.c.segx_bn_ws_floats(B, C, S))

    # ---- r04: two-launch training BatchNorm, squeeze-excite in 2 + 3 launches -----------------------
    def plane_chunks(self, S):
        return int(self.c.segx_plane_chunks(S))

    def bn_pool_chunks(self, B, S, auto_stats):
        return int(self.c.segx_bn_pool_chunks(B, S, 1 if auto_stats else 0))

    def bn_parts_floats(self, B, C, S=0):
        return int(self.c.segx_bn_parts_floats(B, C, S))

    def bn_stats_local(self, X, part, ws, B, C, S):
        self._call('segx_bn_stats_local', X, X, part, ws, B, C, S)

    def bn_act_fwd2(self, X, parts, nparts, mean, var, run_mean, run_var, momentum, w, b, Y, psum, resid, dc_p, seed, offset, B, C, S, eps, act, y_bs=0):
        """y_bs: batch stride of Y in floats when Y is a channel slice of a wider tensor (its (sample, channel) planes contiguous); 0 = dense"""
        if y_bs:
            self._chk_t(Y)
            assert Y.device == X.device and Y.dtype == torch.float32 and Y.stride(1) == S and Y.stride(0) == y_bs and Y[0].is_contiguous()
            Y = Y.data_ptr()                                     # a strided view: the pointer and the stride are the contract
        self._call('segx_bn_act_fwd2', X, X, parts, nparts, mean, var, run_mean, run_var, momentum, w, b, Y, psum, resid, float(dc_p), seed, offset, B, C, S, eps, act,
                   0 if parts is None else parts.numel(), int(y_bs))

    def bn_act_bwd2(self, dY, X, mean, var, w, b, dX, dw, db, ws, B, C, S, eps, act, training, gate=None, dpool=None, inv_S=0.0, dc_p=0.0, seed=0, offset=0, dy_bs=0):
        """dy_bs: batch stride of dY in floats when dY is a channel slice of a wider tensor (its (sample, channel) planes contiguous); 0 = dense"""
        self._chk_t(dY, X, mean, var, w, b, dX, dw, db, ws, gate, dpool)
        for t in (X, mean, var, w, b, dX, dw, db, ws):
            assert t.is_contiguous()
        assert dy_bs or dY.is_contiguous()
        rc = self.c.segx_bn_act_bwd2(_ptr(dY), _ptr(X), _ptr(mean), _ptr(var), _ptr(w), _ptr(b), _ptr(dX), _ptr(dw), _ptr(db), _ptr(ws), B, C, S, eps, act, training,
                                     _ptr(gate), _ptr(dpool), float(inv_S), float(dc_p), seed, offset, int(dy_bs), ws.numel(), self.stream(X))
        self.check(rc, 'segx_bn_act_bwd2')

    def se_fwd2(self, psum, nch, inv_S, W1, b1, W2, b2, Wproj, p, hpre, gate, Wb, B, C, Cs, M):
        self._call('segx_se_fwd2', gate, psum, nch, inv_S, W1, b1, W2, b2, Wproj, p, hpre, gate, Wb, B, C, Cs, M)

    def se_ws2(self, B, C, Cs):
        return int(self.c.segx_se_ws2_floats(B, C, Cs))

    def se_bwd2(self, dWb, Wproj, dgate, gate, hpre, p, W1, W2, inv_S, dpool, dW1, db1, dW2, db2, dWproj, ws, B, C, Cs, M):
        self._call('segx_se_bwd2', gate, dWb, Wproj, dgate, gate, hpre, p, W1, W2, inv_S, dpool, dW1, db1, dW2, db2, dWproj, ws, B, C, Cs, M)

    def dwconv2d_fwd(self, X, W, Y, B, C, H, Wd, OH, OW, k, stride, pt, pl):
        self._call('segx_dwconv2d_fwd', X, X, W, Y, B, C, H, Wd, OH, OW, k, stride, pt, pl)

    def dwconv2d_bwd_data(self, dY, W, dX, B, C, H, Wd, OH, OW, k, stride, pt, pl):
        self._call('segx_dwconv2d_bwd_data', dY, dY, W, dX, B, C, H, Wd, OH, OW, k, stride, pt, pl)

    def dwconv2d_bwd_weight_direct(self, dY, X, dW, B, C, H, Wd, OH, OW, k, stride, pt, pl):
        """-> True when dW was computed in one launch (segx_dwconv2d_bwd_weight_direct), False when the shape needs the two-stage form"""
        self._chk_t(dY, X, dW)
        rc = self.c.segx_dwconv2d_bwd_weight_direct(_ptr(dY), _ptr(X), _ptr(dW), B, C, H, Wd, OH, OW, k, stride, pt, pl, self.stream(X))
        if rc < 0:
            self.check(rc, 'segx_dwconv2d_bwd_weight_direct')
        return rc == 1

    def dwconv2d_bwd_fused_rows(self, H, W, OH, OW, k, stride, pad_t, pad_l):
        return int(self.c.segx_dwconv2d_bwd_fused_rows(H, W, OH, OW, k, stride, pad_t, pad_l))

    def dwconv2d_bwd_fused(self, dY, X, W, dX, part, B, C, H, Wd, OH, OW, k, stride, pad_t, pad_l):
        self._call('segx_dwconv2d_bwd_fused', dY, dY, X, W, dX, part, B, C, H, Wd, OH, OW, k, stride, pad_t, pad_l)

    def dwconv2d_wgrad_rows(self, OH, OW):
        return int(self.c.segx_dwconv2d_wgrad_rows(OH, OW))

    def dwconv2d_bwd_weight(self, dY, X, part, B, C, H, Wd, OH, OW, k, stride, pt, pl):
        self._call('segx_dwconv2d_bwd_weight', dY, dY, X, part, B, C, H, Wd, OH, OW, k, stride, pt, pl)

    def plane_scale(self, X, gate, Y, planes, S):
        self._call('segx_plane_scale', X, X, gate, Y, planes, S)

    def bn_act_bwd_reduce(self, dY, X, mean, var, w, b, dw, db, ws, B, C, S, eps, act, gate=None, dpool=None, inv_S=0.0, dc_p=0.0, seed=0, offset=0):
        self._call('segx_bn_act_bwd_reduce', X, dY, X, mean, var, w, b, dw, db, ws, B, C, S, eps, act, gate, dpool, float(inv_S), float(dc_p), seed, offset)

    def bn_act_bwd_apply(self, dY, X, mean, var, w, b, sdw, sdb, dX, B, C, S, eps, act, inv_n, gate=None, dpool=None, inv_S=0.0, dc_p=0.0, seed=0, offset=0):
        self._call('segx_bn_act_bwd_apply', X, dY, X, mean, var, w, b, sdw, sdb, dX, B, C, S, eps, act, inv_n, gate, dpool, float(inv_S), float(dc_p), seed, offset)

    def plane_bias_add(self, X, bias, Y, planes, C, S):
        self._call('segx_plane_bias_add', X, X, bias, Y, planes, C, S)

    def plane_dot(self, A, B, out, planes, S):
        self._call('segx_plane_dot', A, A, B, out, planes, S)

    # ---- feature-pyramid kernels (fpn.hip) ----------------------------------------------------------
    def interp_fwd_axis2(self, inp, base, out, outer, n1_in, n1_out, n2_in, n2_out, inner):
        self._call('segx_interp_linear_fwd_axis2', inp, inp, base, out, outer, n1_in, n1_out, n2_in, n2_out, inner)

    def interp_bwd_axis2(self, dout, din, outer, n1_out, n1_in, n2_out, n2_in, inner):
        self._call('segx_interp_linear_bwd_axis2', dout, dout, din, outer, n1_out, n1_in, n2_out, n2_in, inner)

    def gn_ws(self, B, C, G):
        return int(self.c.segx_gn_ws_floats(B, C, G))

    def groupnorm_fwd(self, X, w, b, Y, mean, rstd, ws, B, C, G, S, eps):
        self._call('segx_groupnorm_fwd', X, X, w, b, Y, mean, rstd, ws, B, C, G, S, eps)

    def groupnorm_bwd(self, dY, X, w, mean, rstd, dX, dw, db, ws, B, C, G, S, plane_dx_sums=None):
        self._call('segx_groupnorm_bwd', X, dY, X, w, mean, rstd, dX, dw, db, ws, B, C, G, S, plane_dx_sums)

    def interp_gn_nparts(self, per4, cpg):
        return int(self.c.segx_interp_gn_nparts(int(per4), int(cpg)))

    def interp_fwd_axis2_gn(self, x, base, out, outer, n1_in, n1_out, n2_in, n2_out, inner, cpg, parts, nparts):
        self._call('segx_interp_linear_fwd_axis2_gn', x, x, base, out, outer, n1_in, n1_out, n2_in, n2_out, inner, cpg, parts, nparts)

    def groupnorm_stats_parts(self, parts, nparts, mean, rstd, BG, eps):
        self._call('segx_groupnorm_stats_parts', parts, parts, nparts, mean, rstd, BG, eps)

    def gn_fold_bwd(self, Gd, X, mean, rstd, A, Bc, dX, plane_sums, ws, B, C, G, S):
        self._call('segx_gn_fold_bwd', X, Gd, X, mean, rstd, A, Bc, dX, plane_sums, ws, B, C, G, S)

    def gn_fold_bwd_proj(self, dOut, Wb, NC, X, mean, rstd, A, Bc, dX, plane_sums, ws, B, C, G, S):
        self._call('segx_gn_fold_bwd_proj', X, dOut, Wb, NC, X, mean, rstd, A, Bc, dX, plane_sums, ws, B, C, G, S)

    def bridge_mask(self, X, Wb, bb, out, B, Cb, C3, H, W, D, kd, kh, kw):
        self._call('segx_bridge_mask', X, X, Wb, bb, out, B, Cb, C3, H, W, D, kd, kh, kw)

    def groupnorm_fwd_parts(self, X, parts, nparts, w, b, Y, mean, rstd, B, C, G, S, eps):
        self._call('segx_groupnorm_fwd_parts', X, X, parts, nparts, w, b, Y, mean, rstd, B, C, G, S, eps)

    def interp_fwd(self, inp, base, out, planes, d, h, w, D, H, W):
        self._call('segx_interp_linear_fwd', inp, inp, base, out, planes, d, h, w, D, H, W)

    def interp_fwd_axis(self, inp, base, out, outer, n_in, n_out, inner, src_scale=0.0):
        self._call('segx_interp_linear_fwd_axis', inp, inp, base, out, outer, n_in, n_out, inner, float(src_scale))

    def resized_crop3d(self, X, Y, planes, geom):
        self._chk_t(X, Y)
        self.check(self.c.segx_resized_crop3d(_ptr(X), _ptr(Y), planes, self._geom(geom), self.stream(Y)), 'segx_resized_crop3d')

    # ---- augmentation (augment.hip) -------------------------------------------------------------
    def axis_gather(self, X, Y, planes, geom):
        self._chk_t(X, Y)
        self.check(self.c.segx_axis_gather(_ptr(X), _ptr(Y), planes, self._geom(geom), self.stream(Y)), 'segx_axis_gather')

    def pixel_shuffle2(self, X, Y, planes, h, w, inverse):
        self._call('segx_pixel_shuffle2', X, X, Y, planes, h, w, int(inverse))

    def add_noise(self, X, noise, Y, mu, sigma, nonzero_only, seed, offset):
        self._call('segx_add_noise', X, X, noise, Y, X.numel(), float(mu), float(sigma), int(nonzero_only), seed, offset)

    def resize2d(self, X, Y, planes, h, w, H, W, mode, quantize):
        self._call('segx_resize2d', X, X, Y, planes, h, w, H, W, mode, int(quantize))

    def color_blend(self, X, Y, B, HW, mode, factor, pivot, quantize):
        self._call('segx_color_blend', X, X, Y, B, HW, mode, factor, pivot, int(quantize))

    def gray_mean(self, X, mean, B, HW, quantize):
        ws = torch.empty(int(self.c.segx_gray_mean_ws_floats(B, HW)), dtype=torch.float32, device=X.device)
        self._call('segx_gray_mean', X, X, mean, ws, B, HW, int(quantize))

    def normalize(self, X, Y, B, C, HW, scale, mean, std):
        self._call('segx_normalize', X, X, Y, B, C, HW, float(scale), mean, std)

    def interp_bwd(self, dout, din, planes, d, h, w, D, H, W):
        self._call('segx_interp_linear_bwd', dout, dout, din, planes, d, h, w, D, H, W)

    # ---- implicit-GEMM conv3d + max-pool (conv3d.hip) -----------------------------------------------
    @staticmethod
    def _geom(vals):
        return (ctypes.c_int32 * len(vals))(*[int(v) for v in vals])

    def _timed(self, ref, flops, tag, fn):
        if self.gemm_prof is not None and ref.is_cuda:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            self.x6_launches()
            e0.record(); rc = fn(); e1.record()
            self.gemm_prof.append((e0, e1, flops, tag, self.x6_launches() > 0))
            return rc
        return fn()

    def avgpool2_fwd(self, X, Y, planes, H, W):
        self._call('segx_avgpool2_fwd', X, X, Y, planes, H, W)

    def avgpool2_bwd(self, dY, dX, planes, H, W):
        self._call('segx_avgpool2_bwd', dY, dY, dX, planes, H, W)

    def transpose(self, X, Y, batch, R, C):
        self._call('segx_transpose', X, X, Y, batch, R, C)

    # ---- evaluation path (infer.hip) --------------------------------------------------------------
    def window_accum(self, scores, acc, cnt, B, C, geom):
        self._chk_t(scores, acc, cnt)
        g = (c_i * 12)(*[int(v) for v in geom])
        self.check(self.c.segx_window_accum(_ptr(scores), _ptr(acc), _ptr(cnt), B, C, g, self.stream(acc)), 'segx_window_accum')

    def harden_segmap(self, acc, cnt, soft, hard, B, C, S, mode, T=0.5):
        self._call('segx_harden_segmap', hard, acc, cnt, soft, hard, B, C, S, mode, T)

    def dice_sums(self, pred, gt, planes, S):
        n = int(self.c.segx_dice_ws_floats(planes, S))
        part = torch.empty(n, dtype=torch.float32, device=pred.device)
        self._call('segx_dice_sums', pred, pred, gt, part, planes, S)
        chunks = n // (3 * planes)
        out = torch.empty(planes * 3, dtype=torch.float32, device=pred.device)
        ws = torch.empty(self.colreduce_ws(chunks, planes * 3, 1), dtype=torch.float32, device=pred.device)
        self.colsum(part, out, ws, chunks, planes * 3)
        return out.view(planes, 3)

    def conv3d_splitk(self, B, Cout, geom, wgrad):
        return int(self.c.segx_conv3d_splitk(B, Cout, self._geom(geom), 1 if wgrad else 0))

    def conv3d_fwd(self, X, W, Y, B, Cout, geom, splitk=1, ws=None, packed=False, x_bs=0, y_bs=0):
        """x_bs / y_bs (packed only): X / Y are channel slices of wider tensors whose samples lie that many floats apart (0 = dense)."""
        self._chk_t(X, W, Y, ws)
        P = geom[4] * geom[5] * geom[6]; K = geom[0] * geom[7] * geom[8] * geom[9]
        if x_bs or y_bs:
            assert packed
            call = lambda: self.c.segx_conv3d_fwd_packed_bs(_ptr(X), _ptr(W), _ptr(Y), B, Cout, self._geom(geom), splitk, _ptr(ws), int(x_bs), int(y_bs), self.stream(Y))
        else:
            fn = self.c.segx_conv3d_fwd_packed if packed else self.c.segx_conv3d_fwd
            call = lambda: fn(_ptr(X), _ptr(W), _ptr(Y), B, Cout, self._geom(geom), splitk, _ptr(ws), self.stream(Y))
        rc = self._timed(Y, 2.0 * B * Cout * P * K, ('conv3d_fwd', Cout, P, K, B, splitk), call)
        self.check(rc, 'segx_conv3d_fwd')

    # ---- 3 x 3 x 3 stride-1 'same' convolutions with an LDS-resident halo (conv3d_halo.hip, r06) ----
    def conv3d_halo_ok(self, B, Cout, geom):
        return bool(self.c.segx_conv3d_halo_ok(B, Cout, self._geom(geom)))

    def conv3d_halo_pack(self, W, O, C, mode):
        """W [Cout][Cin][3][3][3] -> the pre-split (three bf16 planes) filter bank of the halo kernels; mode 0 forward (O = Cout, C = Cin), 1 backward-data (O = Cin, C = Cout)"""
        Wq = torch.empty(int(self.c.segx_conv3d_halo_wq_floats(O, C)), dtype=torch.float32, device=W.device)
        self._call('segx_conv3d_halo_pack', Wq, W, Wq, O, C, mode)
        return Wq

    def conv3d_halo_fwd(self, X, Wq, Y, B, Cout, geom, x_bs=0, y_bs=0, mtile=0):
        self._chk_t(X, Wq, Y)
        P = geom[4] * geom[5] * geom[6]; K = geom[0] * 27
        call = lambda: self.c.segx_conv3d_halo_fwd(_ptr(X), _ptr(Wq), _ptr(Y), B, Cout, self._geom(geom), int(x_bs), int(y_bs), int(mtile), self.stream(Y))
        rc = self._timed(Y, 2.0 * B * Cout * P * K, ('conv3d_halo_fwd', Cout, P, K, B, 1), call)
        self.check(rc, 'segx_conv3d_halo_fwd')

    def conv3d_halo_wgrad_ok(self, B, Cout, geom):
        return bool(self.c.segx_conv3d_halo_wgrad_ok(B, Cout, self._geom(geom)))

    def conv3d_halo_wgrad(self, dY, X, dW, B, Cout, geom, dy_bs=0, x_bs=0):
        """dW [Cout, Cin, 3, 3, 3] (summed over the batch) of a 3 x 3 x 3 stride-1 'same' convolution, resident-halo form"""
        self._chk_t(dY, X, dW)
        ws = torch.empty(int(self.c.segx_conv3d_halo_wgrad_ws_floats(B, Cout, self._geom(geom))), dtype=torch.float32, device=dW.device)
        P = geom[4] * geom[5] * geom[6]; N = geom[0] * 27
        call = lambda: self.c.segx_conv3d_halo_wgrad(_ptr(dY), _ptr(X), _ptr(dW), _ptr(ws), B, Cout, self._geom(geom), int(dy_bs), int(x_bs), self.stream(dW))
        rc = self._timed(dW, 2.0 * B * Cout * P * N, ('conv3d_halo_wgrad', Cout, N, P, B, 1), call)
        self.check(rc, 'segx_conv3d_halo_wgrad')

    def conv3d_bwd_data_direct(self, dY, W, dX, B, Cout, geom):
        wt = torch.empty(W.numel(), dtype=torch.float32, device=W.device)
        self._chk_t(dY, W, dX)
        rc = self.c.segx_conv3d_bwd_data_direct(_ptr(dY), _ptr(W), _ptr(dX), _ptr(wt), B, Cout, self._geom(geom), self.stream(dX))
        self.check(rc, 'segx_conv3d_bwd_data_direct')

    def stem_compose_fwd(self, Ws, Wb, bb, Wc, O, C3, Cb, Cc, T):
        self._call('segx_stem_compose_fwd', Wc, Ws, Wb, bb, Wc, O, C3, Cb, Cc, T)

    def stem_compose_bwd(self, dWc, Ws, Wb, bb, dWs, dWb, dbb, O, C3, Cb, Cc, T):
        self._call('segx_stem_compose_bwd', dWc, dWc, Ws, Wb, bb, dWs, dWb, dbb, O, C3, Cb, Cc, T)

    def conv2d_stem_fwd(self, X, W, Y, B, Cin, Cout, H, Wd, OH, OW, K, stride, pt, pl):
        self._call('segx_conv2d_stem_fwd', X, X, W, Y, B, Cin, Cout, H, Wd, OH, OW, K, stride, pt, pl)

    def conv2d_stem_im2col(self, X, Xcol, B, Cin, H, Wd, OH, OW, K, stride, pt, pl, rows):
        self._call('segx_conv2d_stem_im2col', X, X, Xcol, B, Cin, H, Wd, OH, OW, K, stride, pt, pl, rows)

    def stem_s2d_input(self, X, Y, B, Cb, H, W, D, U):
        self._call('segx_stem_s2d_input', X, X, Y, B, Cb, H, W, D, U)

    def bridge_input(self, X, Y, B, Cb, Cc, H, W, D):
        self._call('segx_bridge_input', X, X, Y, B, Cb, Cc, H, W, D)

    def nonzero_mask(self, X, out, B, C, D, H, W, kd, kh, kw):
        self._call('segx_nonzero_mask', X, X, out, B, C, D, H, W, kd, kh, kw)

    def label_nhot(self, labels, out, B, Cin, S, mode):
        self._call('segx_label_nhot', out, labels, out, B, Cin, S, mode)

    def conv3d_pack_weights(self, W, Wp, O, C, KV, mode):
        self._call('segx_conv3d_pack_weights', W, W, Wp, O, C, KV, mode)

    def conv3d_flip_weights(self, W, Wt, Cout, Cin, KV):
        self._call('segx_conv3d_flip_weights', W, W, Wt, Cout, Cin, KV)

    def conv3d_bwd_weight(self, dY, X, dWb, B, Cout, geom, splitk, ws, packed=False, dy_bs=0, x_bs=0):
        self._chk_t(dY, X, dWb, ws)
        g = [int(v) for v in geom]
        N, P = g[0] * g[7] * g[8] * g[9], g[4] * g[5] * g[6]
        if dy_bs or x_bs:
            assert packed
            call = lambda: self.c.segx_conv3d_bwd_weight_packed_bs(_ptr(dY), _ptr(X), _ptr(dWb), B, Cout, self._geom(geom), splitk, _ptr(ws), int(dy_bs), int(x_bs),
                                                                   self.stream(dWb))
        else:
            fn = self.c.segx_conv3d_bwd_weight_packed if packed else self.c.segx_conv3d_bwd_weight
            call = lambda: fn(_ptr(dY), _ptr(X), _ptr(dWb), B, Cout, self._geom(geom), splitk, _ptr(ws), self.stream(dWb))
        rc = self._timed(dWb, 2.0 * B * Cout * P * N, ('conv3d_wgrad', Cout, N, P, B, splitk), call)
        self.check(rc, 'segx_conv3d_bwd_weight')

    def conv3d_unpack_wgrad(self, dWp, dW, Cout, Cin, KV):
        self._call('segx_conv3d_unpack_wgrad', dWp, dWp, dW, Cout, Cin, KV)

    def maxpool3d_fwd(self, X, Y, arg, planes, geom):
        self._chk_t(X, Y, arg)
        rc = self.c.segx_maxpool3d_fwd(_ptr(X), _ptr(Y), _ptr(arg), planes, self._geom(geom), self.stream(Y))
        self.check(rc, 'segx_maxpool3d_fwd')

    def maxpool3d_bwd(self, dY, arg, dX, planes, geom, addend=None):
        self._chk_t(dY, arg, dX, addend)
        assert addend is None or (addend.is_contiguous() and addend.shape == dX.shape)
        rc = self.c.segx_maxpool3d_bwd(_ptr(dY), _ptr(arg), _ptr(dX), planes, self._geom(geom), _ptr(addend), self.stream(dX))
        self.check(rc, 'segx_maxpool3d_bwd')

    def dropout(self, x, y, n, p, seed, offset):
        self._call('segx_dropout', x, x, y, n, float(p), seed, offset)

    def interp_bwd_axis(self, dout, din, outer, n_out, n_in, inner, src_scale=0.0):
        self._call('segx_interp_linear_bwd_axis', dout, dout, din, outer, n_out, n_in, inner, float(src_scale))

    def mt_bertadam_step(self, tabs, ntensors, nchunks, chunk, max_global, max_tensor, sched, b1, b2, eps, ws):
        """tabs: dict of device tensors params/grads/m/v (int64 pointer tables), sizes, chunk_tensor, chunk_off,
        chunk_first, active, lr, wd."""
        t = tabs
        self._call('segx_mt_bertadam_step', ws, t['params'], t['grads'], t['m'], t['v'], t['sizes'], t['chunk_tensor'],
                   t['chunk_off'], t['chunk_first'], t['active'], t['lr'], t['wd'], ntensors, nchunks, chunk,
                   max_global, max_tensor, sched, b1, b2, eps, ws)


    def mt_gather(self, src_tab, tabs, chunk_begin, nchunks, chunk):
        """src_tab: device int64 table of this step's gradient addresses (0 = none); tabs['grads'] = the flat slices."""
        self._call('segx_mt_gather', src_tab, src_tab, tabs['grads'], tabs['sizes'], tabs['chunk_tensor'], tabs['chunk_off'],
                   chunk_begin, nchunks, chunk)


# C signatures (include/segx.h): p pointer, i int32, l int64, f float, u uint64; trailing p = stream
_SIGS = {
    'segx_posbias_fwd': 'ppplipffpp', 'segx_posbias_bwd': 'pppplipffp',
    'segx_softmax_fwd': 'ppplifpfuup', 'segx_softmax_bwd': 'pppplifpfuup',
    'segx_layernorm_fwd': 'pppppplifp', 'segx_layernorm_bwd': 'pppppplip',
    'segx_colreduce_ws_floats': 'lli', 'segx_colsum': 'pppllp', 'segx_ln_param_grad': 'ppppppplip',
    'segx_sum': 'plppfp', 'segx_rowsum': 'ppllp',
    'segx_prenorm_fwd': 'pppplfppplliiffuup'.replace('lliif', 'liif'),
    'segx_prenorm_bwd': 'ppppplfpppplii' + 'fuup',
    'segx_prenorm_bwd_all_ws_floats': 'ii', 'segx_prenorm_bwd_all': 'ppppplfpppppppiiifuup',
    'segx_posembed_fwd': 'pppppliifp', 'segx_posembed_bwd': 'ppppppliip',
    'segx_modes_aggr_fwd': 'pppppppiliffuup', 'segx_modes_aggr_bwd': 'ppppppppilifuup',
    'segx_modes_aggr_param_grad': 'pppppppppppilifuup', 'segx_gelu_bwd': 'ppplfuup',
    'segx_gelu_bwd_colsum': 'ppppplifuup', 'segx_gelu_bwd_colsum_ws_floats': 'li', 'segx_modes_aggr_bwd_all_ws_floats': 'ili', 'segx_modes_aggr_bwd_all': 'ppppppppppppilifuup',
    'segx_loss_ws_floats': 'ii', 'segx_seg_loss_fwd': 'ppppppiilfp', 'segx_seg_loss_bwd': 'pppppppiilfp',
    'segx_mt_bertadam_step': 'pppppppppppiiiffffffpp', 'segx_mt_gather': 'pppppiiip',
    'segx_gn_ws_floats': 'iii', 'segx_groupnorm_fwd': 'pppppppiiilfp', 'segx_groupnorm_bwd': 'pppppppppiiilpp',
    'segx_interp_gn_nparts': 'li', 'segx_interp_linear_fwd_axis2_gn': 'pppliiiilipip', 'segx_groupnorm_fwd_parts': 'ppipppppiiilfp',
    'segx_groupnorm_stats_parts': 'pippifp', 'segx_gn_fold_bwd': 'pppppppppiiilp', 'segx_bridge_mask': 'ppppiiiiiiiiip', 'segx_gn_fold_bwd_proj': 'ppippppppppiiilp',
    'segx_interp_linear_fwd': 'pppliiiiiip', 'segx_interp_linear_fwd_axis2': 'pppliiiilp', 'segx_interp_linear_bwd_axis2': 'ppliiiilp', 'segx_interp_linear_bwd': 'ppliiiiiip', 'segx_interp_linear_bwd_axis': 'ppliilfp',
    'segx_axis_gather': 'pplpp', 'segx_pixel_shuffle2': 'ppliiip', 'segx_add_noise': 'ppplffiuup', 'segx_resize2d': 'ppliiiiiip', 'segx_color_blend': 'ppilippip',
    'segx_gray_mean_ws_floats': 'il', 'segx_gray_mean': 'pppilip', 'segx_normalize': 'ppiilfppp',
    'segx_x6_presplit_elems': 'iiii', 'segx_x6_presplit': 'piilliillpp',
    'segx_tune': 'ii', 'segx_tune_get': 'i', 'segx_set_rng_base': 'p', 'segx_rng_advance': 'pup', 'segx_resized_crop3d': 'pplpp', 'segx_stem_compose_fwd': 'ppppiiiiip', 'segx_stem_compose_bwd': 'pppppppiiiiip', 'segx_bridge_input': 'ppiiiiiip', 'segx_stem_s2d_input': 'ppiiiiiip', 'segx_conv2d_stem_fwd': 'pppiiiiiiiiiiip', 'segx_conv2d_stem_im2col': 'ppiiiiiiiiiiip', 'segx_dropout': 'pplfuup', 'segx_avgpool2_fwd': 'ppliip', 'segx_avgpool2_bwd': 'ppliip', 'segx_transpose': 'ppliip', 'segx_interp_linear_fwd_axis': 'pppliilfp', 'segx_window_accum': 'pppiipp', 'segx_harden_segmap': 'ppppiilifp', 'segx_dice_ws_floats': 'll', 'segx_dice_sums': 'pppllp',
    'segx_conv3d_fwd': 'pppiipipp', 'segx_conv3d_fwd_packed': 'pppiipipp', 'segx_conv3d_fwd_packed_bs': 'pppiipipllp', 'segx_conv3d_bwd_weight_packed_bs': 'pppiipipllp', 'segx_conv3d_pack_weights': 'ppiiiip', 'segx_conv3d_splitk': 'iipi', 'segx_conv3d_flip_weights': 'ppiiip', 'segx_conv3d_bwd_weight': 'pppiipipp', 'segx_conv3d_bwd_weight_packed': 'pppiipipp', 'segx_conv3d_unpack_wgrad': 'ppiiip',
    'segx_conv3d_halo_ok': 'iip', 'segx_conv3d_halo_wq_floats': 'ii', 'segx_conv3d_halo_pack': 'ppiiip', 'segx_conv3d_halo_fwd': 'pppiipllip', 'segx_conv3d_halo_wgrad_ok': 'iip', 'segx_conv3d_halo_wgrad_ws_floats': 'iip', 'segx_conv3d_halo_wgrad': 'ppppiipllp',
    'segx_conv3d_bwd_data_direct': 'ppppiipp', 'segx_nonzero_mask': 'ppiiiiiiiip', 'segx_label_nhot': 'ppiilip',
    'segx_maxpool3d_fwd': 'ppplpp', 'segx_maxpool3d_bwd': 'ppplppp',
    'segx_bn_ws_floats': 'iil', 
    'segx_dwconv2d_fwd': 'pppiiiiiiiiiip', 'segx_dwconv2d_bwd_data': 'pppiiiiiiiiiip',
    'segx_dwconv2d_bwd_weight': 'pppiiiiiiiiiip', 'segx_dwconv2d_bwd_weight_direct': 'pppiiiiiiiiiip', 'segx_dwconv2d_wgrad_rows': 'ii', 'segx_dwconv2d_bwd_fused_rows': 'iiiiiiii', 'segx_dwconv2d_bwd_fused': 'pppppiiiiiiiiiip', 'segx_plane_scale': 'pppllp', 'segx_plane_dot': 'pppllp',
     'segx_plane_bias_add': 'ppplilp', 
    'segx_plane_chunks': 'l', 'segx_bn_pool_chunks': 'ili', 'segx_bn_parts_floats': 'iil', 'segx_bn_stats_local': 'pppiilp',
    'segx_bn_act_fwd2': 'ppippppfpppppfuuiilfillp', 'segx_bn_act_bwd2': 'ppppppppppiilfiippffuullp',
    'segx_team_status': 'i', 'segx_team_cap': '', 'segx_occupy': 'iifpp',
    'segx_se_fwd2': 'pifpppppppppiiiip', 'segx_se_ws2_floats': 'iii', 'segx_se_bwd2': 'ppppppppfpppppppiiiip',
    'segx_bn_act_bwd_reduce': 'pppppppppiilfippffuup', 'segx_bn_act_bwd_apply': 'pppppppppiilfifppffuup',
}

_LIB = None
DEFAULT_ENGINE = 'x6'


def lib():
    """The product library (HIP, in-tree).  Fails loudly when it has not been built."""
    global _LIB
    if _LIB is None:
        _LIB = SegxLib(LIB_PATH)
        # tile engine of the process: 'x6' (default: bf16x6, fp32-equivalent) or 'f32' (v_mfma_f32_32x32x2_f32 everywhere); the whole -m gpu
        # parity suite runs on both (tests/test_gpu_model.py, tests/test_gpu_fullshape.py)
        _LIB.set_engine(os.environ.get('SEGX_ENGINE', DEFAULT_ENGINE))
    return _LIB


def use_library(libobj):
    """TEST HOOK: install another build of the same C ABI (the fiber-emulated one from tests/hipemu) so the
    autograd layer can be exercised on CPU tensors.  Never called by product code; pass None to reset."""
    global _LIB
    _LIB = libobj
