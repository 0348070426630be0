"""The rollout replayed as a captured HIP graph (RolloutWorker.use_graph) against the eager rollout: same seeds, same
chips -> identical episodes, statistics and epsilon schedule, over several collect+learn rounds (the graph must see the
weights the learns in between produced, incl. the zero-padded copy of rnn.weight_ih), and for greedy evaluation."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _trainer(use_graph, name='dmfb'):
    from marl_dmfb_amd.common.arguments import make_args
    from marl_dmfb_amd.train import Trainer
    if name == 'dmfb':
        from marl_dmfb_amd.env.dmfb import VecDMFB
        env = VecDMFB(10, 10, 4, fov=9, n_envs=256, seed=11, device='cuda:0')
        kw = {}
    else:
        from marl_dmfb_amd.env.meda import VecMEDA
        env = VecMEDA(30, 30, 4, fov=19, n_envs=64, seed=11, device='cuda:0', version=2)
        kw = dict(name='meda', drop_num=4, width=30, length=30, fov=19)
    torch.manual_seed(5)
    args = make_args(device='cuda:0', n_envs=env.n_envs, batch_size=64, train_time=2, buffer_size=4 * env.n_envs, anneal_steps=20000,
                     use_graph=use_graph, **kw, **env.get_env_info())
    return Trainer(env, args)


@pytest.mark.parametrize('name', ['dmfb', 'meda'])
def test_graph_rollout_equals_eager_rollout(name):
    a, b = _trainer(False, name), _trainer(True, name)
    assert b.rolloutWorker.use_graph and not a.rolloutWorker.use_graph
    b.agents.policy.eval_rnn.load_state_dict(a.agents.policy.eval_rnn.state_dict())
    b.agents.policy.target_rnn.load_state_dict(a.agents.policy.target_rnn.state_dict())
    # building a graph plays one warm-up episode outside the capture (lazy initialisations must not be captured): the chips
    # and the device-side draw counter of the eager side are advanced by the same throw-away episode
    a.agents.policy.init_hidden(1)
    a.rolloutWorker._play(a.rolloutWorker.epsilon.clone(), False, True)
    for rnd in range(3):
        ra = a.rolloutWorker.generate_episode()
        rb = b.rolloutWorker.generate_episode()
        for k in range(4):
            assert torch.equal(ra[k], rb[k]), ('stat', k, rnd)
        for key in ra[4]:
            assert torch.equal(ra[4][key], rb[4][key]), (key, rnd)
        assert float(a.rolloutWorker.epsilon) == float(b.rolloutWorker.epsilon)
        # the same learn on both sides: identical batches (same sampler seed), so the weights stay identical
        for tr, ep in ((a, ra[4]), (b, rb[4])):
            tr.buffer.store_episode(ep)
        for tr in (a, b):
            if tr.buffer.generator is not None:
                tr.buffer.generator.manual_seed(100 + rnd)
            else:
                torch.manual_seed(100 + rnd)
        batch = a.buffer.sample(64)
        if b.buffer.generator is None:
            torch.manual_seed(100 + rnd)
        batch_b = b.buffer.sample(64)
        for key in batch:
            assert torch.equal(torch.as_tensor(batch[key]), torch.as_tensor(batch_b[key])), key
        a.agents.train(batch, rnd)
        b.agents.train(batch_b, rnd)
        for (ka, pa), (kb, pb) in zip(a.agents.policy.eval_rnn.state_dict().items(), b.agents.policy.eval_rnn.state_dict().items()):
            assert torch.equal(pa, pb), ('weights diverged', ka, rnd)
    a.agents.policy.init_hidden(1)
    a.rolloutWorker._play(0.0, True, False)    # the evaluation graph's warm-up episode
    ea = a.rolloutWorker._generate_episode()   # greedy evaluation episode (its own graph)
    eb = b.rolloutWorker._generate_episode()
    for k in range(4):
        assert torch.equal(ea[k], eb[k]), ('eval stat', k)


def test_graph_replay_sees_a_map_injected_after_capture():
    """A transition captured into a HIP graph while the chips' health still followed from the generator (4-bit degrade counts,
    DESIGN.md section 2) must gather the float64 map once one is injected with set_map -- the switch lives in device memory
    (DevPtrs::dflags), not in a by-value kernel argument frozen at capture.  Injected health 0.0 everywhere: no droplet may move
    (`u <= health` never holds, dmfb.py:335), so every chip plays to the step limit and fails; the eager twin agrees."""
    from marl_dmfb_amd.agent.agent import Agents
    from marl_dmfb_amd.common.arguments import make_args
    from marl_dmfb_amd.common.rollout import Evaluator
    from marl_dmfb_amd.env.dmfb import VecDMFB
    outs = []
    for use_graph in (True, False):
        env = VecDMFB(10, 10, 4, fov=9, n_envs=128, seed=3, device='cuda:0', b_degrade=True, per_degrade=1.0)
        torch.manual_seed(2)
        args = make_args(device='cuda:0', n_envs=env.n_envs, **env.get_env_info())
        ev = Evaluator(env, Agents(args), args.episode_limit)
        ev.use_graph = use_graph
        first = ev._generate_episode()                # graph side: warm-up episode + capture + one replay
        if not use_graph:
            first = ev._generate_episode()            # same number of episodes on the eager side
        env.set_map('health', torch.zeros((128, 10, 10), dtype=torch.float64, device='cuda:0'))
        second = ev._generate_episode()
        outs.append((first, second))
        assert int(second[3].sum()) == 0 and int((second[1] != args.episode_limit).sum()) == 0, 'droplets moved on dead electrodes'
    for k in range(4):
        assert torch.equal(outs[0][1][k], outs[1][1][k]), ('stat after set_map', k)


@pytest.mark.parametrize('use_graph', [False, True])
def test_rollout_skipping_finished_chips_equals_the_full_batch_rollout(use_graph):
    """Chips whose episode is over are kept out of the conv front end and the GRU-head kernel (Evaluator.compact_every; the list of
    live chips is rebuilt on the device every k lock-steps).  Episodes, statistics and the epsilon schedule must not change.
    Tasks: every droplet one cell from its goal (or on it), fully random actions, so chips finish at scattered times."""
    from marl_dmfb_amd.env.dmfb import VecDMFB
    E, n = 300, 4
    rng = torch.Generator().manual_seed(3)
    starts = torch.zeros((E, n, 2), dtype=torch.int32)
    starts[:, :, 0] = torch.tensor([1, 4, 7, 4])
    starts[:, :, 1] = torch.tensor([1, 4, 7, 8])
    ends = starts.clone()
    off = torch.randint(0, 3, (E, n), generator=rng)        # 0: on the goal, 1 / 2: one cell away in x / y
    off[:, :3] *= (torch.rand(E, 1, generator=rng) < 0.2).long()   # four chips in five: only the last droplet is off its goal
    ends[:, :, 0] += (off == 1).int()
    ends[:, :, 1] -= (off == 2).int()
    outs = []
    for every in (0, 1, 4):
        tr = _trainer(use_graph)
        env = tr.env
        assert env.n_envs == 256
        env.set_task(starts[:256], ends[:256])
        w = tr.rolloutWorker
        w.reset_fn = env.restart
        w.compact_every = every
        w.live_share = 0.5          # (as Trainer / evaluate report it after a round with early finishers): the live list is in use
        w.epsilon = torch.tensor(1.0, device='cuda')
        res = []
        for _ in range(2):
            r = w.generate_episode()
            res.append((tuple(x.clone() for x in r[:4]), {k: v.clone() for k, v in r[4].items()}, float(w.epsilon)))
        outs.append(res)
        lengths = (~r[4]['padded'][:, :, 0]).sum(1)   # (r[1] is forced to the limit for unsuccessful episodes)
        assert 0.3 < float((lengths < tr.args.episode_limit).float().mean()), 'the scenario must make many chips finish early'
        assert len(torch.unique(lengths)) > 5
    for other in outs[1:]:
        for (sa, ea, epa), (sb, eb, epb) in zip(outs[0], other):
            for k in range(4):
                assert torch.equal(sa[k], sb[k]), ('stat', k)
            for key in ea:
                assert torch.equal(ea[key], eb[key]), key
            assert epa == epb
