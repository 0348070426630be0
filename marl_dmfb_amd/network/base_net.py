"""Per-agent Q-network of the reference (network/base_net.py:23-71), batched over (envs x agents).

Same modules and `state_dict` keys (conv1, conv2[, conv3], mlp1, rnn, fc1) so reference
checkpoints load unchanged.  Two additions for the vectorised loop:
  * `features()` / `recurrent()` split the forward so that VDN.learn can run the non-recurrent
    part (convs + vector MLP) ONCE over all T time steps instead of T times;
  * `forward_obs()` takes the int8 observation rows the HIP env writes plus the last-action
    one-hot, so the rollout never materialises the concatenated float input on the host.
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as f


# fov -> conv stack, each entry (in_is_image, stride).  Reference network/base_net.py:23-33; for
# fov 19 the last two entries are the SAME module object (tied weights), reproduced below.
_CONV_PLAN = {5: [(True, 1)], 7: [(True, 1), (False, 1)], 9: [(True, 1), (False, 1)], 11: [(True, 1), (False, 1)],
              13: [(True, 1), (False, 1)], 19: [(True, 2), (False, 1), (False, 1)]}


class _ConvGemm(torch.autograd.Function):
    """y = cols @ W^T + b for an im2col matrix with millions of rows and a few dozen columns.
    Only the backward differs from addmm: the weight gradient reduces over the huge M dimension, which
    a single skinny GEMM does on a handful of workgroups; here it is split into S independent chunks
    (batched GEMM) that are summed afterwards, and no gradient is formed for `cols` when it does not
    need one (first layer: the observation)."""

    @staticmethod
    def forward(ctx, cols, weight2d, bias):
        ctx.save_for_backward(cols, weight2d)
        return torch.addmm(bias, cols, weight2d.t())

    @staticmethod
    def backward(ctx, g):
        cols, w = ctx.saved_tensors
        g = g.contiguous()
        gcols = g @ w if ctx.needs_input_grad[0] else None
        M, K = cols.shape
        C = g.shape[1]
        S = 1
        while S < 512 and M % (S * 2) == 0 and M // (S * 2) >= 1024:
            S *= 2
        gw = torch.bmm(g.view(S, M // S, C).transpose(1, 2), cols.view(S, M // S, K)).sum(0)
        return gcols, gw, _colsum(g)


N_PART = 256  # partial gradient vectors of the conv backward kernel (one per persistent workgroup)


def _conv9_backward(obs_i8, x, g, w1c, b1c, w2c, od):
    """include/crnn_ops.h: crnn_conv9_backward -> flat dW2 | db2 | dW1 | db1."""
    import ctypes as C
    from .. import _lib
    lib = _lib.crnn_ops()
    vp = C.c_void_p
    if g.stride(1) != 1:
        g = g.contiguous()
    R = obs_i8.shape[0]
    tot = torch.empty(od * od * 9 + od + od * 27 + od, dtype=torch.float32, device=g.device)
    part = torch.empty((N_PART, lib.crnn_conv9_backward_parts(od)), dtype=torch.float32, device=g.device)
    rc = lib.crnn_conv9_backward(vp(obs_i8.data_ptr()), obs_i8.stride(0), R, vp(x.data_ptr()), x.stride(0), vp(g.data_ptr()),
                                 g.stride(0), vp(w1c.data_ptr()), vp(b1c.data_ptr()), vp(w2c.data_ptr()), od, vp(part.data_ptr()),
                                 N_PART, vp(tot.data_ptr()), vp(torch.cuda.current_stream(g.device).cuda_stream))
    if rc != 0:
        raise RuntimeError('crnn_conv9_backward failed: %d (hip %d)' % (rc, lib.crnn_last_hip_error()))
    return tot


def _mlp_branch_backward(obs_i8, dir_off, onehot_i8, x, g, col0):
    """(dW [10][2 + A], db [10]) of the vector branch relu(mlp1([dir, last action])) from the gradient / output columns
    col0 .. col0+9 of the GRU input rows (include/crnn_ops.h: crnn_mlp_backward; two launches)."""
    import ctypes as C
    from .. import _lib
    lib = _lib.crnn_ops()
    vp = C.c_void_p
    A = onehot_i8.shape[1]
    if g.stride(1) != 1:
        g = g.contiguous()
    g_w = torch.empty((10, 2 + A), dtype=torch.float32, device=g.device)
    g_b = torch.empty((10,), dtype=torch.float32, device=g.device)
    part = torch.empty((lib.crnn_mlp_backward_parts(),), dtype=torch.float32, device=g.device)
    rc = lib.crnn_mlp_backward(vp(obs_i8.data_ptr()), obs_i8.stride(0), dir_off, vp(onehot_i8.data_ptr()), A, obs_i8.shape[0],
                               vp(x.data_ptr()), x.stride(0), vp(g.data_ptr()), g.stride(0), col0, vp(part.data_ptr()),
                               vp(g_w.data_ptr()), vp(g_b.data_ptr()), vp(torch.cuda.current_stream(g.device).cuda_stream))
    if rc != 0:
        raise RuntimeError('crnn_mlp_backward failed: %d (hip %d)' % (rc, lib.crnn_last_hip_error()))
    return g_w, g_b


class _ConvFront9(torch.autograd.Function):
    """conv1+ReLU+conv2+ReLU of int8 observation rows for fov 9 through the hand-written HIP kernels
    (include/crnn_ops.h: crnn_conv9_forward / crnn_conv9_backward), with gradients for the four parameter tensors.
    Used for the eval network inside VDN.learn on the GPU when the vector branch cannot be fused (see _Front9Train)."""

    @staticmethod
    def forward(ctx, obs_i8, w1, b1, w2, b2):
        import ctypes as C
        from .. import _lib
        lib = _lib.crnn_ops()
        vp = C.c_void_p
        obs_i8 = obs_i8.contiguous()
        R, od = obs_i8.shape[0], w1.shape[0]
        out = torch.empty((R, od * 25), dtype=torch.float32, device=obs_i8.device)
        w1c, b1c, w2c, b2c = (t.detach().contiguous() for t in (w1, b1, w2, b2))
        rc = lib.crnn_conv9_forward(vp(obs_i8.data_ptr()), obs_i8.stride(0), R, vp(w1c.data_ptr()), vp(b1c.data_ptr()),
                                    vp(w2c.data_ptr()), vp(b2c.data_ptr()), od, vp(out.data_ptr()), out.stride(0),
                                    vp(torch.cuda.current_stream(obs_i8.device).cuda_stream))
        if rc != 0:
            raise RuntimeError('crnn_conv9_forward failed: %d (hip %d)' % (rc, lib.crnn_last_hip_error()))
        ctx.save_for_backward(obs_i8, out, w1c, b1c, w2c)   # nothing extra is saved: the backward recomputes conv1
        ctx.shapes = (w1.shape, w2.shape)
        return out

    @staticmethod
    def backward(ctx, g):
        obs_i8, out, w1c, b1c, w2c = ctx.saved_tensors
        (s1, s2) = ctx.shapes
        od = s1[0]
        n2 = od * od * 9
        tot = _conv9_backward(obs_i8, out, g, w1c, b1c, w2c, od)
        return None, tot[n2 + od:n2 + od + od * 27].view(s1), tot[n2 + od + od * 27:], tot[:n2].view(s2), tot[n2:n2 + od]


class _Front9Train(torch.autograd.Function):
    """The whole GRU input row x = cat([conv features, relu(mlp1([dir, last action]))]) of the eval network in ONE
    launch (crnn_front9_forward) with a hand-written backward: crnn_conv9_backward for the four conv tensors (reading
    the row-strided gradient in place, recomputing conv1) and crnn_mlp_backward for mlp1."""

    @staticmethod
    def forward(ctx, obs_i8, onehot_i8, w1, b1, w2, b2, mlp_w, mlp_b, cols):
        import ctypes as C
        from .. import _lib
        lib = _lib.crnn_ops()
        vp = C.c_void_p
        obs_i8, onehot_i8 = obs_i8.contiguous(), onehot_i8.contiguous()
        R, od, A = obs_i8.shape[0], w1.shape[0], onehot_i8.shape[1]
        x = torch.empty((R, cols), dtype=torch.float32, device=obs_i8.device)  # cols > od*25+10: zero tail (GEMM-friendly K)
        w1c, b1c, w2c, b2c, mwc, mbc = (t.detach().contiguous() for t in (w1, b1, w2, b2, mlp_w, mlp_b))
        rc = lib.crnn_front9_forward(vp(obs_i8.data_ptr()), obs_i8.stride(0), vp(onehot_i8.data_ptr()), A, R, vp(w1c.data_ptr()),
                                     vp(b1c.data_ptr()), vp(w2c.data_ptr()), vp(b2c.data_ptr()), vp(mwc.data_ptr()),
                                     vp(mbc.data_ptr()), od, vp(x.data_ptr()), x.stride(0), cols,
                                     vp(torch.cuda.current_stream(obs_i8.device).cuda_stream))
        if rc != 0:
            raise RuntimeError('crnn_front9_forward failed: %d (hip %d)' % (rc, lib.crnn_last_hip_error()))
        ctx.save_for_backward(obs_i8, onehot_i8, x, w1c, b1c, w2c)
        ctx.shapes = (w1.shape, w2.shape)
        return x

    @staticmethod
    def backward(ctx, g):
        obs_i8, onehot_i8, x, w1c, b1c, w2c = ctx.saved_tensors
        (s1, s2) = ctx.shapes
        od = s1[0]
        n2 = od * od * 9
        tot = _conv9_backward(obs_i8, x, g, w1c, b1c, w2c, od)
        g_mw, g_mb = _mlp_branch_backward(obs_i8, 243, onehot_i8, x, g, od * 25)
        return (None, None, tot[n2 + od:n2 + od + od * 27].view(s1), tot[n2 + od + od * 27:], tot[:n2].view(s2), tot[n2:n2 + od],
                g_mw, g_mb, None)


class _Front19Train(torch.autograd.Function):
    """The GRU input row of the eval network for fov 19 (MEDA): x = cat([conv features, relu(mlp1([dir, last action]))]) in ONE
    launch (crnn_front19_forward: stride-2 conv1, then the tied conv3 twice) with a hand-written backward
    (crnn_conv19_backward: recomputes a1 / a2 on the matrix cores, transposed convolutions in gather form, both applications of
    conv3 add into one weight gradient) and crnn_mlp_backward for mlp1."""

    @staticmethod
    def forward(ctx, obs_i8, onehot_i8, w1, b1, w3, b3, mlp_w, mlp_b, cols):
        import ctypes as C
        from .. import _lib
        lib = _lib.crnn_ops()
        vp = C.c_void_p
        obs_i8, onehot_i8 = obs_i8.contiguous(), onehot_i8.contiguous()
        R, od, A = obs_i8.shape[0], w1.shape[0], onehot_i8.shape[1]
        x = torch.empty((R, cols), dtype=torch.float32, device=obs_i8.device)
        w1c, b1c, w3c, b3c, mwc, mbc = (t.detach().contiguous() for t in (w1, b1, w3, b3, mlp_w, mlp_b))
        rc = lib.crnn_front19_forward(vp(obs_i8.data_ptr()), obs_i8.stride(0), vp(onehot_i8.data_ptr()), A, R, vp(w1c.data_ptr()),
                                      vp(b1c.data_ptr()), vp(w3c.data_ptr()), vp(b3c.data_ptr()), vp(mwc.data_ptr()),
                                      vp(mbc.data_ptr()), od, vp(x.data_ptr()), x.stride(0), cols,
                                      vp(torch.cuda.current_stream(obs_i8.device).cuda_stream))
        if rc != 0:
            raise RuntimeError('crnn_front19_forward failed: %d (hip %d)' % (rc, lib.crnn_last_hip_error()))
        ctx.save_for_backward(obs_i8, onehot_i8, x, w1c, b1c, w3c, b3c)
        ctx.shapes = (w1.shape, w3.shape)
        return x

    @staticmethod
    def backward(ctx, g):
        import ctypes as C
        from .. import _lib
        lib = _lib.crnn_ops()
        vp = C.c_void_p
        obs_i8, onehot_i8, x, w1c, b1c, w3c, b3c = ctx.saved_tensors
        (s1, s3) = ctx.shapes
        od = s1[0]
        if g.stride(1) != 1:
            g = g.contiguous()
        R = obs_i8.shape[0]
        n3 = od * od * 9
        tot = torch.empty(n3 + od + od * 27 + od, dtype=torch.float32, device=g.device)
        part = torch.empty((N_PART, lib.crnn_conv19_backward_parts(od)), dtype=torch.float32, device=g.device)
        rc = lib.crnn_conv19_backward(vp(obs_i8.data_ptr()), obs_i8.stride(0), R, vp(x.data_ptr()), x.stride(0), vp(g.data_ptr()),
                                      g.stride(0), vp(w1c.data_ptr()), vp(b1c.data_ptr()), vp(w3c.data_ptr()), vp(b3c.data_ptr()), od,
                                      vp(part.data_ptr()), N_PART, vp(tot.data_ptr()), vp(torch.cuda.current_stream(g.device).cuda_stream))
        if rc != 0:
            raise RuntimeError('crnn_conv19_backward failed: %d (hip %d)' % (rc, lib.crnn_last_hip_error()))
        g_mw, g_mb = _mlp_branch_backward(obs_i8, 1083, onehot_i8, x, g, od * 25)
        return (None, None, tot[n3 + od:n3 + od + od * 27].view(s1), tot[n3 + od + od * 27:], tot[:n3].view(s3), tot[n3:n3 + od],
                g_mw, g_mb, None)


class _LinearSplitK(torch.autograd.Function):
    """x @ W^T (+ b) for very tall x: the weight gradient g^T @ x has few outputs and a reduction over all the rows,
    which is done as a split-K batched GEMM (`_wgrad_splitk`)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return torch.addmm(bias, x, weight.t()) if bias is not None else torch.matmul(x, weight.t())

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        g = g.contiguous()
        gx = torch.matmul(g, w) if ctx.needs_input_grad[0] else None
        return gx, _wgrad_splitk(g, x), (_colsum(g) if ctx.has_bias else None)


def _colsum(t2d):
    """Column sums of a very tall matrix (bias gradients): two stages, so that the first one has thousands of
    independent outputs instead of a handful (a single at::sum over 81 920 x 10 takes 55 us, this 12 us)."""
    M = t2d.shape[0]
    for cand in (256, 128, 64, 32):
        if M % cand == 0 and M // cand >= 64:
            return t2d.reshape(cand, M // cand, -1).sum(1).sum(0)
    return t2d.sum(0)


def _wgrad_splitk(g2d, x2d):
    """g2d^T @ x2d for tall operands (M rows, few output elements): the reduction over M is split into S
    independent chunks (batched GEMM) so that the whole chip works on it, then the S partials are added."""
    M = g2d.shape[0]
    S = 1
    for cand in (64, 32, 16, 8, 4, 2):
        if M % cand == 0 and M // cand >= 512:
            S = cand
            break
    if S == 1:
        return torch.matmul(g2d.t(), x2d)
    return torch.bmm(g2d.view(S, M // S, -1).transpose(1, 2), x2d.view(S, M // S, -1)).sum(0)


class _GRUSeq(torch.autograd.Function):
    """GRU cell unrolled over a whole sequence as ONE autograd node (GPU only).

    forward: per step `h @ W_hh^T` + the fused gate kernel (`_thnn_fused_gru_cell`, what nn.GRUCell
    dispatches to).  backward: per step the fused gate backward + one small GEMM for the hidden
    gradient; the weight gradient of W_hh and both bias gradients are formed ONCE from the stacked
    gate gradients instead of T accumulations of tiny per-step results."""

    @staticmethod
    def forward(ctx, igates, h0, w_hh, b_ih, b_hh):
        T = igates.shape[0]
        w_hh_t = w_hh.t()
        h, hs, wss = h0, [], []
        for t in range(T):
            h, ws = torch.ops.aten._thnn_fused_gru_cell(igates[t], torch.matmul(h, w_hh_t), h, b_ih, b_hh)
            hs.append(h)
            wss.append(ws)
        out = torch.stack(hs, dim=0)
        ctx.save_for_backward(out, h0, w_hh, *wss)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        out, h0, w_hh, *wss = ctx.saved_tensors
        T, R, H = out.shape
        d_ig = torch.empty((T, R, 3 * H), dtype=out.dtype, device=out.device)
        d_hg = torch.empty_like(d_ig)
        gh = None
        for t in range(T - 1, -1, -1):
            g = grad_out[t] if gh is None else grad_out[t] + gh
            gi, ghg, ghx, _, _ = torch.ops.aten._thnn_fused_gru_cell_backward(g.contiguous(), wss[t], False)
            d_ig[t] = gi
            d_hg[t] = ghg
            gh = torch.addmm(ghx, ghg, w_hh)
        h_prev = torch.cat([h0.unsqueeze(0), out[:-1]], dim=0)
        d_w_hh = torch.matmul(d_hg.view(T * R, 3 * H).t(), h_prev.view(T * R, H))
        return d_ig, gh, d_w_hh, d_ig.sum(dim=(0, 1)), d_hg.sum(dim=(0, 1))


class _GRUSeqHip(torch.autograd.Function):
    """The same node as `_GRUSeq`, but the whole time loop is ONE launch each way (include/crnn_ops.h
    gru_seq_forward / gru_seq_backward, csrc/gru_ops.hip): W_hh lives in registers, h in LDS, a workgroup
    owns 8 rows for all T steps.  Hidden size 128 only (every shipped config)."""

    @staticmethod
    def _lib():
        from .. import _lib
        return _lib.crnn_ops()

    @staticmethod
    def run_forward(igates, h0, w_hh, b_ih, b_hh, save):
        import ctypes as C
        lib = _GRUSeqHip._lib()
        T, R, G = igates.shape
        H = G // 3
        igates, h0, w_hh = igates.contiguous(), h0.contiguous(), w_hh.contiguous()
        hs = torch.empty((T, R, H), dtype=torch.float32, device=igates.device)
        gates = torch.empty((T, R, 4 * H), dtype=torch.float32, device=igates.device) if save else None
        vp = C.c_void_p
        rc = lib.gru_seq_forward(vp(igates.data_ptr()), vp(h0.data_ptr()), vp(w_hh.data_ptr()), vp(b_ih.data_ptr()),
                                 vp(b_hh.data_ptr()), T, R, H, vp(hs.data_ptr()),
                                 vp(gates.data_ptr()) if save else None,
                                 vp(torch.cuda.current_stream(igates.device).cuda_stream))
        if rc != 0:
            raise RuntimeError('gru_seq_forward failed: %d (hip %d)' % (rc, lib.gru_last_hip_error()))
        return hs, gates

    @staticmethod
    def forward(ctx, igates, h0, w_hh, b_ih, b_hh):
        hs, gates = _GRUSeqHip.run_forward(igates.detach(), h0.detach(), w_hh.detach(), b_ih.detach(), b_hh.detach(), True)
        ctx.save_for_backward(hs, gates, h0, w_hh)
        return hs

    @staticmethod
    def backward(ctx, grad_out):
        import ctypes as C
        hs, gates, h0, w_hh = ctx.saved_tensors
        lib = _GRUSeqHip._lib()
        T, R, H = hs.shape
        grad_out = grad_out.contiguous()
        h0c, w = h0.contiguous(), w_hh.contiguous()
        d_ig = torch.empty((T, R, 3 * H), dtype=torch.float32, device=hs.device)
        d_hg = torch.empty_like(d_ig)
        d_h0 = torch.empty_like(h0c)
        bias_part = torch.empty((lib.gru_seq_row_blocks(R), 6 * H), dtype=torch.float32, device=hs.device)
        vp = C.c_void_p
        rc = lib.gru_seq_backward(vp(grad_out.data_ptr()), vp(gates.data_ptr()), vp(hs.data_ptr()), vp(h0c.data_ptr()),
                                  vp(w.data_ptr()), T, R, H, vp(d_ig.data_ptr()), vp(d_hg.data_ptr()), vp(d_h0.data_ptr()),
                                  vp(bias_part.data_ptr()), vp(torch.cuda.current_stream(hs.device).cuda_stream))
        if rc != 0:
            raise RuntimeError('gru_seq_backward failed: %d (hip %d)' % (rc, lib.gru_last_hip_error()))
        # dW_hh = sum_t d_hg[t]^T h_{t-1}: h_0 separately, the rest straight from the saved hs (no concatenated copy)
        d_w_hh = torch.matmul(d_hg[0].t(), h0c)
        if T > 1:
            d_w_hh = d_w_hh + _wgrad_splitk(d_hg[1:].reshape((T - 1) * R, 3 * H), hs[:-1].reshape((T - 1) * R, H))
        d_b = bias_part.sum(0)  # per-row-block column sums from the kernel: db_ih | db_hh
        return d_ig, d_h0, d_w_hh, d_b[:3 * H], d_b[3 * H:]


class _GRUSeqHipPacked(torch.autograd.Function):
    """`_GRUSeqHip` for sequences of different lengths stored without their padded steps (include/crnn_ops.h:
    gru_seq_forward_packed): rows sorted by length, step t holds the first step_rows[t] rows, steps back to back.  igates is
    (V_pad, 3H) with V = sum(step_rows) real rows; rows V .. V_pad-1 (GEMM-friendly padding) are zero in every output."""

    @staticmethod
    def _steps(step_rows):
        import ctypes as C
        return (C.c_int32 * len(step_rows))(*[int(v) for v in step_rows])

    @staticmethod
    def run_forward(igates, h0, w_hh, b_ih, b_hh, step_rows, save):
        import ctypes as C
        lib = _GRUSeqHip._lib()
        Vp, G = igates.shape
        H, T, R, V = G // 3, len(step_rows), h0.shape[0], int(sum(step_rows))
        igates, h0, w_hh = igates.contiguous(), h0.contiguous(), w_hh.contiguous()
        hs = torch.empty((Vp, H), dtype=torch.float32, device=igates.device)
        hs[V:].zero_()
        gates = torch.empty((Vp, 4 * H), dtype=torch.float32, device=igates.device) if save else None
        vp = C.c_void_p
        rc = lib.gru_seq_forward_packed(vp(igates.data_ptr()), vp(h0.data_ptr()), vp(w_hh.data_ptr()), vp(b_ih.data_ptr()),
                                        vp(b_hh.data_ptr()), T, R, H, _GRUSeqHipPacked._steps(step_rows), vp(hs.data_ptr()),
                                        vp(gates.data_ptr()) if save else None,
                                        vp(torch.cuda.current_stream(igates.device).cuda_stream))
        if rc != 0:
            raise RuntimeError('gru_seq_forward_packed failed: %d (hip %d)' % (rc, lib.gru_last_hip_error()))
        return hs, gates

    @staticmethod
    def forward(ctx, igates, h0, w_hh, b_ih, b_hh, step_rows):
        hs, gates = _GRUSeqHipPacked.run_forward(igates.detach(), h0.detach(), w_hh.detach(), b_ih.detach(), b_hh.detach(), step_rows, True)
        ctx.save_for_backward(hs, gates, h0, w_hh)
        ctx.step_rows = list(step_rows)
        return hs

    @staticmethod
    def backward(ctx, grad_out):
        import ctypes as C
        hs, gates, h0, w_hh = ctx.saved_tensors
        lib = _GRUSeqHip._lib()
        step_rows = ctx.step_rows
        Vp, H = hs.shape
        T, R, V = len(step_rows), h0.shape[0], int(sum(step_rows))
        grad_out = grad_out.contiguous()
        h0c, w = h0.contiguous(), w_hh.contiguous()
        d_ig = torch.empty((Vp, 3 * H), dtype=torch.float32, device=hs.device)
        d_hg = torch.empty_like(d_ig)
        h_prev = torch.empty((Vp, H), dtype=torch.float32, device=hs.device)
        for t in (d_ig, d_hg, h_prev):
            t[V:].zero_()
        d_h0 = torch.zeros_like(h0c)
        bias_part = torch.empty((lib.gru_seq_row_blocks(R), 6 * H), dtype=torch.float32, device=hs.device)
        vp = C.c_void_p
        rc = lib.gru_seq_backward_packed(vp(grad_out.data_ptr()), vp(gates.data_ptr()), vp(hs.data_ptr()), vp(h0c.data_ptr()),
                                         vp(w.data_ptr()), T, R, H, _GRUSeqHipPacked._steps(step_rows), vp(d_ig.data_ptr()),
                                         vp(d_hg.data_ptr()), vp(d_h0.data_ptr()), vp(bias_part.data_ptr()), vp(h_prev.data_ptr()),
                                         vp(torch.cuda.current_stream(hs.device).cuda_stream))
        if rc != 0:
            raise RuntimeError('gru_seq_backward_packed failed: %d (hip %d)' % (rc, lib.gru_last_hip_error()))
        d_w_hh = _wgrad_splitk(d_hg, h_prev)   # sum over every running (t, row) of d_hgates^T h_{t-1}
        d_b = bias_part.sum(0)
        return d_ig, d_h0, d_w_hh, d_b[:3 * H], d_b[3 * H:], None


class _GRUSeqPairPacked(torch.autograd.Function):
    """The packed GRU recurrence of TWO networks over the same batch in one launch (include/crnn_ops.h:
    gru_seq_forward_packed_pair): network a with gradients (the eval net of VDN.learn: gates saved, backward =
    gru_seq_backward_packed), network b without (the target net).  Zero initial state.  Returns (hs_a, hs_b)."""

    @staticmethod
    def forward(ctx, ig_a, w_hh_a, b_ih_a, b_hh_a, ig_b, w_hh_b, b_ih_b, b_hh_b, step_rows, R):
        import ctypes as C
        lib = _GRUSeqHip._lib()
        Vp, G = ig_a.shape
        Hd, T, V = G // 3, len(step_rows), int(sum(step_rows))
        save = any(ctx.needs_input_grad[:4])
        ta = [t.detach().contiguous() for t in (ig_a, w_hh_a, b_ih_a, b_hh_a)]
        tb = [t.detach().contiguous() for t in (ig_b, w_hh_b, b_ih_b, b_hh_b)]
        hs_a = torch.empty((Vp, Hd), dtype=torch.float32, device=ig_a.device)
        hs_b = torch.empty((Vp, Hd), dtype=torch.float32, device=ig_a.device)
        hs_a[V:].zero_()
        hs_b[V:].zero_()
        gates = torch.empty((Vp, 4 * Hd), dtype=torch.float32, device=ig_a.device) if save else None
        vp = C.c_void_p
        rc = lib.gru_seq_forward_packed_pair(vp(ta[0].data_ptr()), None, vp(ta[1].data_ptr()), vp(ta[2].data_ptr()), vp(ta[3].data_ptr()),
                                             vp(hs_a.data_ptr()), vp(gates.data_ptr()) if save else None,
                                             vp(tb[0].data_ptr()), None, vp(tb[1].data_ptr()), vp(tb[2].data_ptr()), vp(tb[3].data_ptr()),
                                             vp(hs_b.data_ptr()), None, T, R, Hd, _GRUSeqHipPacked._steps(step_rows),
                                             vp(torch.cuda.current_stream(ig_a.device).cuda_stream))
        if rc != 0:
            raise RuntimeError('gru_seq_forward_packed_pair failed: %d (hip %d)' % (rc, lib.gru_last_hip_error()))
        if save:
            h0 = torch.zeros((R, Hd), dtype=torch.float32, device=ig_a.device)
            ctx.save_for_backward(hs_a, gates, h0, ta[1])
            ctx.step_rows = list(step_rows)
        ctx.mark_non_differentiable(hs_b)
        return hs_a, hs_b

    @staticmethod
    def backward(ctx, grad_a, _grad_b):
        d_ig, _d_h0, d_w_hh, d_b_ih, d_b_hh, _ = _GRUSeqHipPacked.backward(ctx, grad_a)
        return d_ig, d_w_hh, d_b_ih, d_b_hh, None, None, None, None, None, None


def gru_sequence(igates, h0, w_hh, b_ih, b_hh, impl='hip'):
    """hs (T, R, H) of the GRU recurrence given the input-side pre-activations of all steps."""
    if impl == 'hip' and h0.shape[-1] == 128 and igates.dtype == torch.float32:
        if torch.is_grad_enabled() and any(t.requires_grad for t in (igates, h0, w_hh, b_ih, b_hh)):
            return _GRUSeqHip.apply(igates, h0, w_hh, b_ih, b_hh)
        return _GRUSeqHip.run_forward(igates, h0, w_hh, b_ih, b_hh, False)[0]
    return _GRUSeq.apply(igates, h0, w_hh, b_ih, b_hh)


def conv_str(fov, id=3, od=32):
    stack, shared = [], None
    for from_image, stride in _CONV_PLAN[fov]:
        if from_image:
            stack.append(nn.Conv2d(id, od, kernel_size=3, stride=stride))
        else:
            if shared is None:
                shared = nn.Conv2d(od, od, kernel_size=3, stride=stride)
            stack.append(shared)
    return stack


class CRNN(nn.Module):
    def __init__(self, args):
        super().__init__()
        id = args.obs_shape[0]
        od = args.hyper_hidden_dim
        self.input_dim = tuple(args.obs_shape)
        self.rnn_hidden_dim = args.rnn_hidden_dim
        self.n_actions = args.n_actions
        self.convs = conv_str(args.fov, id, od)
        size = args.fov
        for i, conv in enumerate(self.convs, 1):
            self.add_module('conv{}'.format(i), conv)
            size = int((size + 2 * conv.padding[0] - conv.dilation[0] * (conv.kernel_size[0] - 1) - 1) // conv.stride[0] + 1)
        self.out = size * size * od
        self.n_pixel = self.input_dim[-1] - self.input_dim[-2]
        self.mlp1 = nn.Linear(args.obs_shape[-2] + args.n_actions, 10)
        self.rnn = nn.GRUCell(self.out + 10, args.rnn_hidden_dim)
        self.fc1 = nn.Linear(args.rnn_hidden_dim, args.n_actions)
        self.conv_impl = getattr(args, 'conv_impl', 'gemm')  # 'gemm' (GPU default) or 'conv2d'

    # -- non-recurrent part: rows may be (envs*agents) or (episodes*T*agents)
    def _conv_stack_gemm(self, pixel):
        """The conv stack as im2col + ONE large GEMM per layer, activations kept channels-last.
        On the GPU this replaces MIOpen's convolution (whose solver search / kernel compilation for
        these 9x9 images costs minutes on a fresh box) by plain rocBLAS/hipBLASLt GEMMs with
        M = rows x output pixels.  Same arithmetic, different summation order."""
        x = pixel.permute(0, 2, 3, 1)  # (R, H, W, C) view
        for conv in self.convs:
            k, s = conv.kernel_size[0], conv.stride[0]
            win = x.unfold(1, k, s).unfold(2, k, s)          # (R, H', W', C, k, k) view
            R, Ho, Wo = win.shape[0], win.shape[1], win.shape[2]
            cols = win.reshape(R * Ho * Wo, -1)              # im2col, K ordered (c, kh, kw) like conv.weight
            y = _ConvGemm.apply(cols, conv.weight.view(conv.out_channels, -1), conv.bias)
            x = f.relu(y).view(R, Ho, Wo, conv.out_channels)
        return x.permute(0, 3, 1, 2).reshape(x.shape[0], self.out)  # back to the reference's (c, h, w) order

    def features_split(self, pixel, vec):
        pixel = pixel.reshape((-1,) + self.input_dim[:3])
        if pixel.is_cuda and self.conv_impl == 'gemm':
            pixel = self._conv_stack_gemm(pixel)
        else:
            for conv in self.convs:
                pixel = f.relu(conv(pixel))
            pixel = pixel.reshape((-1, self.out))
        vec = f.relu(self.mlp1(vec))
        return torch.cat([pixel, vec], dim=1)

    def features(self, inputs):
        pixel, vec = torch.split(inputs, [self.n_pixel, self.n_actions + self.input_dim[-2]], dim=1)
        return self.features_split(pixel, vec)

    def recurrent(self, x, hidden_state):
        h_in = hidden_state.reshape(-1, self.rnn_hidden_dim)
        h = self.rnn(x, h_in)
        return self.fc1(h), h

    def recurrent_seq(self, x_seq, h0):
        """GRU cell + head over a whole sequence x_seq (T, R, F) from h0 (R, H): returns (q (T, R, A), h_T).
        On the GPU the input projection x @ W_ih^T of ALL steps is one GEMM and the head runs once
        over the stacked hidden states; the recurrence itself is one HIP launch for all T steps
        (`_GRUSeqHip`, hidden 128; `gru_impl='aten'` selects the per-step `_thnn_fused_gru_cell` path,
        the kernel nn.GRUCell dispatches to)."""
        T, R = x_seq.shape[0], x_seq.shape[1]
        h = h0.reshape(-1, self.rnn_hidden_dim)
        hs = []
        if x_seq.is_cuda:
            # rows from the HIP front end come zero-padded to `padded_cols()` (K = 640 / 832 for the GEMM)
            w_ih = self.weight_ih_padded() if x_seq.shape[-1] == self.padded_cols() != self.rnn.weight_ih.shape[1] else self.rnn.weight_ih
            igates = _LinearSplitK.apply(x_seq.reshape(T * R, -1), w_ih, None).view(T, R, -1)
            hseq = gru_sequence(igates, h, self.rnn.weight_hh, self.rnn.bias_ih, self.rnn.bias_hh,
                                getattr(self, 'gru_impl', 'hip'))
            q = _LinearSplitK.apply(hseq.view(T * R, -1), self.fc1.weight, self.fc1.bias).view(T, R, -1)
            return q, hseq[-1]
        else:
            for t in range(T):
                h = self.rnn(x_seq[t], h)
                hs.append(h)
        q = self.fc1(torch.stack(hs, dim=0).view(T * R, -1)).view(T, R, -1)
        return q, h

    def recurrent_seq_packed(self, x, step_rows, R):
        """`recurrent_seq` on PACKED rows (GPU): x (V_pad, F) holds, step after step, the GRU inputs of the rows still running
        (`step_rows[t]` = the first so many of the R length-sorted rows); returns q (V_pad, A) in the same layout.  Zero initial
        hidden state (policy/vdn.py:198-203)."""
        w_ih = self.weight_ih_padded() if x.shape[-1] == self.padded_cols() != self.rnn.weight_ih.shape[1] else self.rnn.weight_ih
        igates = _LinearSplitK.apply(x, w_ih, None)
        h0 = torch.zeros((R, self.rnn_hidden_dim), dtype=torch.float32, device=x.device)
        args = (igates, h0, self.rnn.weight_hh, self.rnn.bias_ih, self.rnn.bias_hh)
        if torch.is_grad_enabled() and any(t.requires_grad for t in args):
            hs = _GRUSeqHipPacked.apply(*args, step_rows)
        else:
            hs = _GRUSeqHipPacked.run_forward(*args, step_rows, False)[0]
        return _LinearSplitK.apply(hs, self.fc1.weight, self.fc1.bias)

    @staticmethod
    def recurrent_seq_packed_pair(net_a, x_a, net_b, x_b, step_rows, R):
        """`recurrent_seq_packed` of two networks over the same packed batch (the eval net `net_a`, with gradients, and the target net
        `net_b` under no_grad) with ONE launch for the two recurrences (include/crnn_ops.h: gru_seq_forward_packed_pair).
        Returns (q_a, q_b), each (V_pad, A)."""
        def w_ih(net, x, grad):
            if x.shape[-1] == net.padded_cols() != net.rnn.weight_ih.shape[1]:
                return net.weight_ih_padded() if grad else net.refresh_padded()
            return net.rnn.weight_ih
        ig_a = _LinearSplitK.apply(x_a, w_ih(net_a, x_a, True), None)
        with torch.no_grad():
            ig_b = torch.matmul(x_b, w_ih(net_b, x_b, False).t())
        hs_a, hs_b = _GRUSeqPairPacked.apply(ig_a, net_a.rnn.weight_hh, net_a.rnn.bias_ih, net_a.rnn.bias_hh,
                                             ig_b, net_b.rnn.weight_hh.detach(), net_b.rnn.bias_ih.detach(), net_b.rnn.bias_hh.detach(),
                                             step_rows, R)
        q_a = _LinearSplitK.apply(hs_a, net_a.fc1.weight, net_a.fc1.bias)
        with torch.no_grad():
            q_b = torch.addmm(net_b.fc1.bias, hs_b, net_b.fc1.weight.t())
        return q_a, q_b

    def forward(self, inputs, hidden_state):
        """Reference signature: inputs (R, obs+n_actions) float32, hidden (R, H) -> (q, h)."""
        return self.recurrent(self.features(inputs), hidden_state)

    def _pixel_features_hip(self, obs_i8):
        """conv1+ReLU+conv2+ReLU of the int8 observation rows through the hand-written HIP kernel
        (include/crnn_ops.h, csrc/crnn_ops.hip); inference only."""
        import ctypes as C
        from .. import _lib
        lib = _lib.crnn_ops()
        obs_i8 = obs_i8.contiguous()
        R = obs_i8.shape[0]
        out = torch.empty((R, self.out), dtype=torch.float32, device=obs_i8.device)
        c1, c2 = self.convs[0], self.convs[1]
        stream = C.c_void_p(torch.cuda.current_stream(obs_i8.device).cuda_stream)
        if self._hip_geometry() == 19:  # pixel features only: no vector branch (NULL mlp pointers)
            rc = lib.crnn_front19_forward(C.c_void_p(obs_i8.data_ptr()), obs_i8.stride(0), None, 0, R,
                                          C.c_void_p(c1.weight.data_ptr()), C.c_void_p(c1.bias.data_ptr()),
                                          C.c_void_p(c2.weight.data_ptr()), C.c_void_p(c2.bias.data_ptr()), None, None,
                                          c1.out_channels, C.c_void_p(out.data_ptr()), out.stride(0), 0, stream)
        else:
            rc = lib.crnn_conv9_forward(C.c_void_p(obs_i8.data_ptr()), obs_i8.stride(0), R,
                                        C.c_void_p(c1.weight.data_ptr()), C.c_void_p(c1.bias.data_ptr()),
                                        C.c_void_p(c2.weight.data_ptr()), C.c_void_p(c2.bias.data_ptr()),
                                        c1.out_channels, C.c_void_p(out.data_ptr()), out.stride(0), stream)
        if rc != 0:
            raise RuntimeError('crnn_conv9_forward failed: %d (hip %d)' % (rc, lib.crnn_last_hip_error()))
        return out

    def padded_cols(self):
        """GRU input width rounded up to a multiple of 64 (include/crnn_ops.h: crnn_front_padded_cols)."""
        return (self.out + 10 + 63) // 64 * 64

    def weight_ih_padded(self):
        """rnn.weight_ih with zero columns up to `padded_cols()` (a fresh tensor; inside the autograd graph when
        gradients are on: the pad's gradient is sliced away)."""
        w = self.rnn.weight_ih
        return f.pad(w, (0, self.padded_cols() - w.shape[1]))

    def refresh_padded(self):
        """The same, copied IN PLACE into a persistent buffer: the rollout calls this once per episode (weights do not
        change inside one) and hands the buffer to `act_gates`; a captured rollout graph re-runs the copy on every
        replay.  No version check on purpose: the fused Adam step does not bump a parameter's `_version`."""
        w = self.rnn.weight_ih.detach()
        buf = getattr(self, '_w_ih_pad', None)
        if buf is None or buf.device != w.device or buf.shape != (w.shape[0], self.padded_cols()):
            buf = self._w_ih_pad = w.new_zeros((w.shape[0], self.padded_cols()))
        buf[:, :w.shape[1]].copy_(w)
        return buf

    def front_features_live(self, obs_i8, onehot_i8, live_chips, n_live, rows_per_chip, out):
        """`_front_features_hip(padded=True)` for the chips listed in `live_chips` (int32, ascending; `n_live` their count, both on
        the device) only: row k * rows_per_chip + a of `out` is the GRU input of row live_chips[k] * rows_per_chip + a of
        obs_i8 / onehot_i8 (include/crnn_ops.h: crnn_front9_forward_live).  Rows of `out` beyond the live ones are left as they are."""
        import ctypes as C
        from .. import _lib
        lib = _lib.crnn_ops()
        c1, c2 = self.convs[0], self.convs[1]
        vp = C.c_void_p
        rc = lib.crnn_front9_forward_live(vp(obs_i8.data_ptr()), obs_i8.stride(0), vp(onehot_i8.data_ptr()), self.n_actions, obs_i8.shape[0],
                                          vp(c1.weight.data_ptr()), vp(c1.bias.data_ptr()), vp(c2.weight.data_ptr()), vp(c2.bias.data_ptr()),
                                          vp(self.mlp1.weight.data_ptr()), vp(self.mlp1.bias.data_ptr()), c1.out_channels,
                                          vp(out.data_ptr()), out.stride(0), out.shape[1], vp(live_chips.data_ptr()), vp(n_live.data_ptr()),
                                          int(rows_per_chip), vp(torch.cuda.current_stream(obs_i8.device).cuda_stream))
        if rc != 0:
            raise RuntimeError('crnn_front9_forward_live failed: %d (hip %d)' % (rc, lib.crnn_last_hip_error()))
        return out

    def _front_features_hip(self, obs_i8, onehot_i8, padded=False):
        """GRU input x = cat([conv features, relu(mlp1([dir, last action]))]) in one HIP launch
        (include/crnn_ops.h: crnn_front9_forward / crnn_front19_forward); inference only.  padded: rows of `padded_cols()`
        floats with a zero tail, for the GEMM against `weight_ih_padded()`."""
        import ctypes as C
        from .. import _lib
        lib = _lib.crnn_ops()
        obs_i8 = obs_i8.contiguous()
        R = obs_i8.shape[0]
        cols = self.padded_cols() if padded else self.out + 10
        out = torch.empty((R, cols), dtype=torch.float32, device=obs_i8.device)
        c1, c2 = self.convs[0], self.convs[1]
        oh = None
        if onehot_i8 is not None:
            onehot_i8 = onehot_i8.to(torch.int8).contiguous()
            oh = C.c_void_p(onehot_i8.data_ptr())
        # fov 19 (MEDA v0_2): stride-2 conv, then the tied conv3 twice (include/crnn_ops.h: crnn_front19_forward)
        fn = lib.crnn_front19_forward if self._hip_geometry() == 19 else lib.crnn_front9_forward
        rc = fn(C.c_void_p(obs_i8.data_ptr()), obs_i8.stride(0), oh, self.n_actions, R,
                                     C.c_void_p(c1.weight.data_ptr()), C.c_void_p(c1.bias.data_ptr()),
                                     C.c_void_p(c2.weight.data_ptr()), C.c_void_p(c2.bias.data_ptr()),
                                     C.c_void_p(self.mlp1.weight.data_ptr()), C.c_void_p(self.mlp1.bias.data_ptr()),
                                     c1.out_channels, C.c_void_p(out.data_ptr()), out.stride(0), cols,
                                     C.c_void_p(torch.cuda.current_stream(obs_i8.device).cuda_stream))
        if rc != 0:
            raise RuntimeError('crnn_front%d_forward failed: %d (hip %d)' % (self._hip_geometry(), rc, lib.crnn_last_hip_error()))
        return out

    def features_obs_train(self, obs_i8, la_rows):
        """GRU input rows for the eval network inside learn (gradients flow to every parameter):
        HIP conv front end with its own backward + the small vector MLP in torch."""
        c1, c2 = self.convs[0], self.convs[1]
        if self._hip_geometry() == 19:  # tied conv3: autograd adds the gradient this node returns for it ONCE (both applications inside)
            return _Front19Train.apply(obs_i8, la_rows.to(torch.int8), c1.weight, c1.bias, c2.weight, c2.bias,
                                       self.mlp1.weight, self.mlp1.bias, self.padded_cols())
        if self.mlp1.in_features == 2 + self.n_actions and self.n_actions <= 16 and self.mlp1.out_features == 10:
            return _Front9Train.apply(obs_i8, la_rows.to(torch.int8), c1.weight, c1.bias, c2.weight, c2.bias,
                                      self.mlp1.weight, self.mlp1.bias, self.padded_cols())
        pix = _ConvFront9.apply(obs_i8, c1.weight, c1.bias, c2.weight, c2.bias)
        vec = torch.cat([obs_i8[:, self.n_pixel:].float(), la_rows.float()], dim=1)
        return torch.cat([pix, f.relu(self.mlp1(vec))], dim=1)

    def _hip_train_ok(self, obs_i8):
        if not (self.conv_impl == 'gemm' and obs_i8.is_cuda and obs_i8.dtype == torch.int8 and torch.is_grad_enabled()
                and self.convs[0].out_channels in (24, 32)):
            return False
        if self._hip_geometry() == 19:   # MEDA: stride-2 conv1 + the tied conv3 twice (crnn_conv19_backward)
            return (self.mlp1.in_features == 2 + self.n_actions and self.n_actions <= 16 and self.mlp1.out_features == 10
                    and os.environ.get('MARL_DMFB_CONV19_BWD', '1') != '0')
        return (self.input_dim[:3] == (3, 9, 9) and len(self.convs) == 2 and self.convs[0] is not self.convs[1])

    def _hip_geometry(self):
        """9 / 19: the conv stack is one of the two the HIP front-end kernels implement (conv_str(9): conv1, conv3;
        conv_str(19): stride-2 conv, then the SAME conv3 module twice); None otherwise."""
        cv = self.convs
        if self.input_dim[:3] == (3, 9, 9) and len(cv) == 2 and cv[0] is not cv[1] and cv[0].stride[0] == 1:
            return 9
        if self.input_dim[:3] == (3, 19, 19) and len(cv) == 3 and cv[1] is cv[2] and cv[0].stride[0] == 2 and cv[1].stride[0] == 1:
            return 19
        return None

    def _hip_conv_ok(self, obs_i8):
        return (self.conv_impl == 'gemm' and obs_i8.is_cuda and obs_i8.dtype == torch.int8 and not torch.is_grad_enabled()
                and self._hip_geometry() is not None and self.convs[0].out_channels in (24, 32)
                and self.convs[0].weight.is_contiguous() and self.convs[1].weight.is_contiguous())

    def act_ok(self, obs_i8):
        """True when a rollout lock-step can use `act_gates` + rollout_gru_head_select (include/rollout_ops.h)."""
        return (self._hip_conv_ok(obs_i8) and self.mlp1.in_features == 2 + self.n_actions and self.n_actions <= 16
                and self.mlp1.out_features == 10)

    def act_gates(self, obs_i8, last_action_onehot, hidden_state, w_ih_padded=None):
        """The GEMM part of one rollout lock-step: x = front end (HIP), then the two GRU projections x W_ih^T and
        h W_hh^T (no bias).  The gate math, fc1 and the epsilon-greedy pick follow in ONE kernel."""
        x = self._front_features_hip(obs_i8, last_action_onehot, padded=True)
        w = self.weight_ih_padded() if w_ih_padded is None else w_ih_padded  # `refresh_padded()` of this episode
        return torch.matmul(x, w.t()), torch.matmul(hidden_state, self.rnn.weight_hh.t())

    def forward_obs(self, obs_i8, last_action_onehot, hidden_state):
        """obs_i8 (R, 3*fov*fov+2) int8 as written by the env kernels; last_action_onehot (R, n_actions)."""
        if self._hip_conv_ok(obs_i8) and self.mlp1.in_features == 2 + self.n_actions and self.n_actions <= 16:
            return self.recurrent(self._front_features_hip(obs_i8, last_action_onehot), hidden_state)
        pixel = obs_i8[:, :self.n_pixel].float()
        vec = torch.cat([obs_i8[:, self.n_pixel:].float(), last_action_onehot.float()], dim=1)
        return self.recurrent(self.features_split(pixel, vec), hidden_state)


class RNN(nn.Module):
    """network/base_net.py:7-21 (args.net == 'rnn')."""

    def __init__(self, input_shape, args):
        super().__init__()
        self.args = args
        self.fc1 = nn.Linear(input_shape, args.rnn_hidden_dim)
        self.rnn = nn.GRUCell(args.rnn_hidden_dim, args.rnn_hidden_dim)
        self.fc2 = nn.Linear(args.rnn_hidden_dim, args.n_actions)

    def features(self, inputs):
        return f.relu(self.fc1(inputs))

    def recurrent(self, x, hidden_state):
        h = self.rnn(x, hidden_state.reshape(-1, self.args.rnn_hidden_dim))
        return self.fc2(h), h

    def forward(self, obs, hidden_state):
        return self.recurrent(self.features(obs), hidden_state)

    def forward_obs(self, obs_i8, last_action_onehot, hidden_state):
        return self.forward(torch.cat([obs_i8.float(), last_action_onehot.float()], dim=1), hidden_state)
