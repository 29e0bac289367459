"""Does the eager launch (246 ctypes launches in ~2.2 ms of host time per step) survive SEVERAL Python ranks on one host?
K worker processes run the bench step concurrently ON THE ONE GPU of the box (the GPU is time-shared between them, so their
step time is not the point): what is read off is each process's HOST time to enqueue a step while K-1 other interpreters
enqueue theirs, and the summed images/s against a single process -- an 8-rank node has 8 such interpreters per host.
usage: host_contention.py [K=4] [steps=30]        (worker mode: host_contention.py --worker <steps>)"""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def worker(steps):
    sys.path[:0] = [ROOT, os.path.join(ROOT, 'ram-dsir_amd')]
    import torch
    from ramdsir import step as S
    import bench as Bn
    bank, mods = S.make_bank('cuda:0', 3, 16, 2, 3)
    Bn.init_weights(bank)
    ts = S.TrainStep(bank, mods, torch.bfloat16, [2, 3, 3], 400, 400, ram='u8')
    ts.wpack.refresh()
    src, trg, lam, mask, _ = Bn.synth_inputs(8, 400, 0, 'cuda:0')
    ts.load_raw(src, trg, lam); ts.load_target(mask)
    for _ in range(5):
        ts.step()
    torch.cuda.synchronize()
    # (a) enqueue cost on an idle queue: one step at a time, host time until the last launch call returns
    one = []
    for _ in range(10):
        torch.cuda.synchronize()
        t0 = time.perf_counter(); ts.step(); one.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
    # (b) throughput: `steps` steps back to back
    t0 = time.perf_counter()
    for _ in range(steps):
        ts.step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print('HCRESULT ' + json.dumps(dict(enqueue_ms_idle=round(1e3 * sorted(one)[len(one) // 2], 3), enqueue_ms_loop=round(1e3 * (t1 - t0) / steps, 3),
                                        ms_per_step=round(1e3 * (t2 - t0) / steps, 3), images_per_s=round(8 * steps / (t2 - t0), 1),
                                        lanes_verified=bool(ts.lanes_verified))), flush=True)


def run(k, steps):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), '--worker', str(steps)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env)
             for _ in range(k)]
    res = []
    for p in procs:
        o, _ = p.communicate(timeout=900)
        lines = [l for l in o.decode().splitlines() if l.startswith('HCRESULT ')]
        if p.returncode != 0 or not lines:
            print(o.decode()[-2000:])
            raise SystemExit('worker failed')
        res.append(json.loads(lines[-1][len('HCRESULT '):]))
    return res


if __name__ == '__main__':
    if len(sys.argv) > 2 and sys.argv[1] == '--worker':
        worker(int(sys.argv[2]))
        sys.exit(0)
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    print('host cores: %d' % (os.cpu_count() or 0))
    for k in (1, K):
        res = run(k, steps)
        print('%d process(es) on one GPU: enqueue (idle queue) %s ms/step, enqueue (in the loop) %s ms/step, step %s ms, '
              'sum %.0f images/s, lanes verified %s' % (k, [r['enqueue_ms_idle'] for r in res], [r['enqueue_ms_loop'] for r in res],
                                                        [r['ms_per_step'] for r in res], sum(r['images_per_s'] for r in res),
                                                        [r['lanes_verified'] for r in res]))
