"""Where does a round of the default loop go?  Wall-clock per phase with a device synchronisation behind each (debug aid)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from marl_dmfb_amd.common.arguments import make_args
from marl_dmfb_amd.env.dmfb import VecDMFB
from marl_dmfb_amd.train import Trainer

E = 4096
env = VecDMFB(10, 10, 4, fov=9, n_envs=E, seed=1234, device='cuda:0')
over = {}
for kv in sys.argv[1:]:
    k, v = kv.split('=')
    over[k] = int(v)
args = make_args(device='cuda:0', n_envs=E, batch_size=512, train_time=4, buffer_size=16384, **over, **env.get_env_info())
torch.manual_seed(1234)
tr = Trainer(env, args)
for _ in range(4):
    tr.collect_and_learn()
torch.cuda.synchronize()
import gc
if os.environ.get('DBG_GC') == 'freeze':
    gc.collect(); gc.freeze()
elif os.environ.get('DBG_GC') == 'disable':
    gc.disable()
print('gc', os.environ.get('DBG_GC'), gc.get_threshold(), gc.get_count(), len(gc.get_objects()))
ms0 = torch.cuda.memory_stats()
t0 = time.perf_counter()
per = []
for _ in range(20):
    t1 = time.perf_counter()
    tr.collect_and_learn()
    per.append(round((time.perf_counter() - t1) * 1e3, 1))
torch.cuda.synchronize()
ms1 = torch.cuda.memory_stats()
print('host ms per collect_and_learn call:', per)
print('segments allocated during the loop: %d, freed: %d, alloc retries: %d, reserved %.2f GB, active peak %.2f GB' % (
    ms1['segment.all.allocated'] - ms0['segment.all.allocated'], ms1['segment.all.freed'] - ms0['segment.all.freed'],
    ms1['num_alloc_retries'] - ms0['num_alloc_retries'], ms1['reserved_bytes.all.current'] / 1e9, ms1['active_bytes.all.peak'] / 1e9))
print('plain loop: %.3f ms per round (stream %s packed %s)' % ((time.perf_counter() - t0) / 20 * 1e3, tr.stream, tr._packed), flush=True)
if tr.stream:
    w, buf, pol = tr.rolloutWorker, tr.buffer, tr.agents.policy
    acc = {}
    def lap(name, t):
        torch.cuda.synchronize()
        acc[name] = acc.get(name, 0.0) + time.perf_counter() - t
    for _ in range(10):
        t = time.perf_counter(); a = w.generate_steps(buf, 40); lap('rollout(sync)', t)
        t = time.perf_counter(); buf.sync_host(a); lap('sync_host', t)
        t = time.perf_counter(); draws = [buf.draw(512) for _ in range(4)]; lap('draw', t)
        for k in range(4):
            t = time.perf_counter()
            if tr._packed:
                pol.learn_packed(buf.buffers, draws[k][0], draws[k][1], tr.trained_times)
            else:
                tr.agents.train(buf.gather(draws[k][0]), tr.trained_times, max_len=int(draws[k][1][0]))
            host = time.perf_counter() - t
            acc['learn host'] = acc.get('learn host', 0.0) + host
            lap('learn(sync)', t)
            tr.trained_times += 1
    print({k: round(v / 10 * 1e3, 3) for k, v in acc.items()}, flush=True)
if tr.stream and tr._packed and os.environ.get('DBG_PROFILE'):
    import cProfile, pstats
    draws = [buf.draw(512) for _ in range(4)]
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for k in range(4):
        pol.learn_packed(buf.buffers, draws[k][0], draws[k][1], tr.trained_times)
    pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats('tottime').print_stats(18)
if tr.stream and os.environ.get('DBG_TIMELINE'):
    torch.cuda.synchronize()
    for rnd in range(3):
        ev = []
        def mark(name):
            e = torch.cuda.Event(enable_timing=True); e.record(); ev.append((name, time.perf_counter(), e))
        base_t = time.perf_counter()
        mark('start')
        a = w.generate_steps(buf, 40); mark('rollout queued')
        buf.sync_host(a); mark('synced')
        draws = [buf.draw(512) for _ in range(4)]
        mark('plans')
        for k in range(4):
            if tr._packed:
                pol.learn_packed(buf.buffers, draws[k][0], draws[k][1], tr.trained_times)
            else:
                tr.agents.train(buf.gather(draws[k][0]), tr.trained_times, max_len=int(draws[k][1][0]))
            mark('learn %d queued' % k)
        torch.cuda.synchronize()
        print('round', rnd, ' | '.join('%s host %.2f gpu %.2f' % (n, (t - base_t) * 1e3, ev[0][2].elapsed_time(e)) for n, t, e in ev), flush=True)
if tr.stream and os.environ.get('DBG_FREE'):
    torch.cuda.synchronize()
    allev = []
    base = torch.cuda.Event(enable_timing=True); base.record(); tb = time.perf_counter()
    for rnd in range(9):
        ev = []
        def mark(name):
            e = torch.cuda.Event(enable_timing=True); e.record(); ev.append((name, time.perf_counter(), e))
        mark('start')
        a = w.generate_steps(buf, 40); mark('roll')
        buf.sync_host(a); mark('sync')
        draws = [buf.draw(512) for _ in range(4)]
        mark('plan')
        for k in range(4):
            if tr._packed:
                pol.learn_packed(buf.buffers, draws[k][0], draws[k][1], tr.trained_times)
            else:
                tr.agents.train(buf.gather(draws[k][0]), tr.trained_times, max_len=int(draws[k][1][0]))
            mark('L%d' % k)
        allev.append(ev)
    torch.cuda.synchronize()
    for rnd, ev in enumerate(allev):
        print('round', rnd, ' | '.join('%s h%.1f g%.1f' % (n, (t - tb) * 1e3, base.elapsed_time(e)) for n, t, e in ev), flush=True)
if tr.stream and os.environ.get('DBG_SAMPLE'):
    import threading, traceback, collections
    main_id = threading.get_ident()
    samples, stop = [], [False]
    def sampler():
        while not stop[0]:
            fr = sys._current_frames().get(main_id)
            if fr is not None:
                st_ = traceback.extract_stack(fr)[-4:]
                samples.append((time.perf_counter(), tuple('%s:%d %s' % (os.path.basename(f.filename), f.lineno, f.name) for f in st_)))
            time.sleep(0.001)
    th = threading.Thread(target=sampler, daemon=True); th.start()
    torch.cuda.synchronize()
    for _ in range(30):
        tr.collect_and_learn()
    torch.cuda.synchronize()
    stop[0] = True; th.join()
    # runs of identical innermost frames
    runs, cur, t_start, t_last = [], None, 0, 0
    for t, stk in samples:
        if stk != cur:
            if cur is not None:
                runs.append((t_last - t_start, cur))
            cur, t_start = stk, t
        t_last = t
    runs.sort(reverse=True)
    for d, stk in runs[:8]:
        print('%.1f ms  %s' % (d * 1e3, ' <- '.join(reversed(stk))), flush=True)
