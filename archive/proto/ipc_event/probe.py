"""Probe (round 4): how many times can an interprocess HIP event be recorded / waited on across two processes on one GPU?
    python3 archive/proto/ipc_event/probe.py [mode]     mode: plain | query | sync16"""
import os
import sys
import torch
import torch.multiprocessing as mp

MODE = sys.argv[1] if len(sys.argv) > 1 else 'plain'


def work(rank, q01, q10, mode):
    torch.cuda.set_device(0)
    ev = torch.cuda.Event(enable_timing=False, interprocess=True)
    ev.record()
    torch.cuda.synchronize()
    (q01 if rank == 0 else q10).put(ev.ipc_handle())
    peer = torch.cuda.Event.from_ipc_handle(torch.device('cuda:0'), (q10 if rank == 0 else q01).get())
    x = torch.zeros(1 << 18, device='cuda')
    st = torch.cuda.current_stream()
    n = 0
    try:
        for i in range(400):
            x.add_(1.0)
            ev.record(st)
            (q01 if rank == 0 else q10).put(i)          # "my record i is issued"
            (q10 if rank == 0 else q01).get()           # peer's record i is issued
            st.wait_event(peer)
            if mode == 'query':
                ev.query()
            if mode == 'sync16' and i % 16 == 15:
                st.synchronize()
            n = i + 1
        torch.cuda.synchronize()
        print('rank %d mode %s: %d exchanges ok' % (rank, mode, n), flush=True)
    except Exception as e:   # noqa: BLE001
        print('rank %d mode %s: failed at exchange %d: %s' % (rank, mode, n, str(e).splitlines()[0]), flush=True)
        (q01 if rank == 0 else q10).put(-1)


if __name__ == '__main__':
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    ctx = mp.get_context('spawn')
    for mode in ([MODE] if MODE != 'all' else ['plain', 'query', 'sync16']):
        q01, q10 = ctx.Queue(), ctx.Queue()
        ps = [ctx.Process(target=work, args=(r, q01, q10, mode)) for r in range(2)]
        for p in ps:
            p.start()
        for p in ps:
            p.join(120)
            if p.is_alive():
                p.terminate()
