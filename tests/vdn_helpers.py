"""Shared by the VDN / rollout golden tests: the deterministic weight formula used when the
goldens were captured from the reference (tools/oracle/gen_vdn_golden.py: det_init)."""
import numpy as np
import torch


def det_init(module, salt=0.0):
    with torch.no_grad():
        for k, (name, p) in enumerate(module.named_parameters()):
            i = torch.arange(p.numel(), dtype=torch.float64)
            scale = 0.08 if p.dim() > 1 else 0.02
            p.copy_((scale * torch.sin(0.37 * i + 1.7 * k + salt)).to(torch.float32).view_as(p))


def learn_golden_check(path, device, rtol, atol, replay_dtypes=False, host_len_bound=False, packed=False, atol_step1=None):
    """replay_dtypes: hand the batch over exactly as ReplayBuffer.sample does on the GPU (r float32, int8 actions, bool flags) and
    REQUIRE the shipped learn path (fused TD block + time-major Q values, policy/vdn.py:_td_fused_ok); otherwise the golden's own
    dtypes (r float64), which take the tensor-op TD block.
    host_len_bound: Agents.train(..., max_len=episode_limit) -- what Trainer.collect_and_learn does by default (the longest episode
    ever stored, handed over from the host): the learn runs over every slot of the batch although the reference trims it to the
    batch's own longest episode (agent/agent.py:51-70); the extra steps are padded in every episode, so the reference's numbers
    must come out all the same.
    packed: VDN.learn_packed -- the golden batch stands in for the replay ring (one slot per episode), the episodes are handed over
    as (slot indices, lengths) sorted by length like ReplayBuffer.draw does, and the padded steps are never computed.
    atol_step1: absolute tolerance (fraction of a tensor's largest reference gradient) of the SECOND learn's gradients when it is not
    `atol` (the caller says why)."""
    from marl_dmfb_amd.agent.agent import Agents
    from marl_dmfb_amd.common.arguments import make_args
    g = np.load(path)
    cfg = [int(v) for v in g['cfg']]
    W, L, n, fov, od, clip = cfg[:6]
    if len(cfg) == 8:   # MEDA network shape (fov 19, 9 actions): tools/oracle/gen_vdn_golden.py:gen_learn_meda
        A, T = cfg[6:]
        args = make_args(name='meda', drop_num=n, width=W, length=L, fov=fov, cuda=(device != 'cpu'), device=device, n_actions=A,
                         n_agents=n, obs_shape=(3, fov, fov, 2, 3 * fov * fov + 2), episode_limit=T)
    else:
        T = 2 * (W + L)
        args = make_args(drop_num=n, width=W, length=L, fov=fov, cuda=(device != 'cpu'), device=device, n_actions=5,
                         n_agents=n, obs_shape=(3, fov, fov, 2, 3 * fov * fov + 2), episode_limit=T)
    assert args.hyper_hidden_dim == od and args.grad_norm_clip == clip
    agents = Agents(args)
    det_init(agents.policy.eval_rnn)
    det_init(agents.policy.target_rnn, salt=0.5)
    keys = ['o', 'u', 'r', 'o_next', 'avail_u', 'avail_u_next', 'u_onehot', 'padded', 'terminated']
    names = [str(x) for x in g['names']]
    assert names == [k for k, _ in agents.policy.eval_rnn.named_parameters()]
    for step in range(2):
        batch = {k: torch.as_tensor(g[k]).to(device) for k in keys}
        batch['padded'] = batch['padded'].bool()
        batch['terminated'] = batch['terminated'].bool()
        if replay_dtypes:
            batch['r'] = batch['r'].float()
            for k in ('u', 'avail_u', 'avail_u_next', 'u_onehot', 'o', 'o_next'):
                batch[k] = batch[k].to(torch.int8)
            assert agents.policy._td_fused_ok(batch), 'the shipped (fused TD) learn path must be the one under test'
        else:
            assert not agents.policy._td_fused_ok(batch)
        if packed:
            assert agents.policy.packed_ok(batch), 'the packed learn path must apply to this golden'
            lens = (~batch['padded'][:, :, 0]).sum(1).cpu().numpy()
            order = np.argsort(-lens, kind='stable')
            agents.policy.learn_packed(batch, order, lens[order], step)
        elif host_len_bound:
            agents.train(batch, step, max_len=batch['o'].shape[1])
        else:
            agents.train(batch, step)
        np.testing.assert_allclose(float(agents.policy.last_grad_norm), g['grad_norm'][step], rtol=rtol)
        for name, p in agents.policy.eval_rnn.named_parameters():
            idx = torch.as_tensor(g['idx/' + name])
            grad = p.grad.detach().reshape(-1).cpu()[idx].numpy()
            w = p.detach().reshape(-1).cpu()[idx].numpy()
            ref_g = g['grad%d/%s' % (step, name)]
            scale = np.abs(ref_g).max() + 1e-12
            np.testing.assert_allclose(grad, ref_g, rtol=rtol, atol=(atol_step1 if step == 1 and atol_step1 is not None else atol) * scale,
                                       err_msg='grad %s step %d' % (name, step))
            np.testing.assert_allclose(w, g['w%d/%s' % (step, name)], rtol=rtol, atol=2e-6, err_msg='w %s step %d' % (name, step))
