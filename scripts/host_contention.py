"""Does the eager launch (246 ctypes launches in ~2.2 ms of host time per step) survive SEVERAL Python ranks on one host?
K worker processes run the bench step concurrently ON THE ONE GPU of the box (the GPU is time-shared between them, so their
step time is not the point): what is read off is each process's HOST time to enqueue a step while K-1 other interpreters
enqueue theirs, and the summed images/s against a single process -- an 8-rank node has 8 such interpreters per host.
usage: host_contention.py [K=4] [steps=30]        (worker mode: host_contention.py --worker <steps>)"""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def worker(steps, start_at=0.0):
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
    # (b) throughput: `steps` steps back to back, all workers from the same wall-clock instant when one is given
    while time.time() < start_at:
        time.sleep(0.001)
    w0 = time.time()
    t0 = time.perf_counter()
    for _ in range(steps):
        ts.step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print('HCRESULT ' + json.dumps(dict(wall_start=w0, wall_end=time.time(), late=bool(start_at and w0 - start_at > 0.05),enqueue_ms_idle=round(1e3 * sorted(one)[len(one) // 2], 3), enqueue_ms_loop=round(1e3 * (t1 - t0) / steps, 3),
                                        ms_per_step=round(1e3 * (t2 - t0) / steps, 3), images_per_s=round(8 * steps / (t2 - t0), 1),
                                        lanes_verified=bool(ts.lanes_verified))), flush=True)


def run(k, steps, lead_s=0.0):
    """lead_s > 0: the workers start their timed loops together, lead_s seconds from now (enough for every interpreter to import torch,
    build the step and warm up); the saturated rate is then all images / (last end - common start)."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    for name in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'RD_FORCE_DDP'):
        env.pop(name, None)
    start_at = time.time() + lead_s if lead_s > 0 else 0.0
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), '--worker', str(steps), repr(start_at)], stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, env=env) for _ in range(k)]
    res = []
    for p in procs:
        o, _ = p.communicate(timeout=900)
        lines = [l for l in o.decode().splitlines() if l.startswith('HCRESULT ')]
        if p.returncode != 0 or not lines:
            print(o.decode()[-2000:])
            raise SystemExit('worker failed')
        res.append(json.loads(lines[-1][len('HCRESULT '):]))
    return res


def saturated_rate(k=4, steps=150, lead_s=30.0):
    """images/s of k processes running the bench step on ONE GPU from a common start: what the kernels deliver when the dependency
    bubbles of one process's step are filled by other processes' kernels (bench.py reports it beside the single-process value)."""
    res = run(k, steps, lead_s)
    if any(r['late'] for r in res):
        return None
    t0, t1 = min(r['wall_start'] for r in res), max(r['wall_end'] for r in res)
    return dict(processes=k, steps_each=steps, images_per_s=round(8 * steps * k / (t1 - t0), 1),
                per_process_ms_per_step=[r['ms_per_step'] for r in res])


if __name__ == '__main__':
    if len(sys.argv) > 2 and sys.argv[1] == '--worker':
        worker(int(sys.argv[2]), float(sys.argv[3]) if len(sys.argv) > 3 else 0.0)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == '--sat':            # common-start saturated rates: host_contention.py --sat 1,2,4,8 [steps]
        for k in [int(x) for x in sys.argv[2].split(',')]:
            print(saturated_rate(k, int(sys.argv[3]) if len(sys.argv) > 3 else 150), flush=True)
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
