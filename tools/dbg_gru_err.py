import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from marl_dmfb_amd.network.base_net import _GRUSeqHipPacked, _GRUSeqPairPacked
H, T, R = 128, 16, 512
torch.manual_seed(0)
cell = torch.nn.GRUCell(H, H).cuda()
with torch.no_grad():
    cell.weight_hh.mul_(3.0)   # larger recurrent weights: the h W_hh^T sums matter
step_rows = [R] * T
V = R * T
ig = torch.randn(V, 3 * H, device='cuda')
h0 = torch.zeros(R, H, device='cuda')
with torch.no_grad():
    hs_v = _GRUSeqHipPacked.run_forward(ig, h0, cell.weight_hh, cell.bias_ih, cell.bias_hh, step_rows, False)[0]
    hs_m, _ = _GRUSeqPairPacked.apply(ig, cell.weight_hh, cell.bias_ih, cell.bias_hh, ig, cell.weight_hh, cell.bias_ih, cell.bias_hh, step_rows, R)
w, bi, bh = cell.weight_hh.detach().double().cpu(), cell.bias_ih.detach().double().cpu(), cell.bias_hh.detach().double().cpu()
ig64 = ig.double().cpu()
h = torch.zeros(R, H, dtype=torch.float64)
ref = []
for t in range(T):
    gi = ig64[t * R:(t + 1) * R] + bi
    gh = h @ w.t() + bh
    r_ = torch.sigmoid(gi[:, :H] + gh[:, :H]); z_ = torch.sigmoid(gi[:, H:2 * H] + gh[:, H:2 * H])
    n_ = torch.tanh(gi[:, 2 * H:] + r_ * gh[:, 2 * H:])
    h = (1 - z_) * n_ + z_ * h
    ref.append(h)
ref = torch.cat(ref, 0)
for name, hs in (('VALU kernel', hs_v), ('MFMA pair kernel', hs_m)):
    e = (hs.double().cpu() - ref).abs()
    print('%-18s max |err| %.3g  rms %.3g' % (name, float(e.max()), float(e.pow(2).mean().sqrt())))
