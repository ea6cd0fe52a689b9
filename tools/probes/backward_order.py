"""Which autograd nodes run in which order during the second-order (R1) phase, iteration after iteration?  Logs (name, sequence_nr)
of every node of the Dreg backward via pre-hooks and compares consecutive iterations.  usage: python tools/probes/backward_order.py"""
import os, sys, copy
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import torch
from test_gpu_train_graph import small_networks, real_batch, DEV
from shgan_amd import losses, train_stage as ts
G, D = small_networks(5)
g0, d0 = copy.deepcopy(G.state_dict()), copy.deepcopy(D.state_dict())
real4 = real_batch(4, 6)
kw = dict(lr=0.002, betas=(0.0, 0.99), eps=1e-8)
LOG = []
orig_backward = torch.Tensor.backward


def walk(root):
    seen, stack, nodes = set(), [root], []
    while stack:
        n = stack.pop()
        if n is None or n in seen:
            continue
        seen.add(n); nodes.append(n)
        stack.extend(f for f, _ in n.next_functions)
    return nodes


def logged_backward(self, *a, **k):
    nodes = walk(self.grad_fn)
    cur = []
    LOG.append(cur)
    import threading
    main = threading.get_ident()
    if len(LOG) <= 5:
        print('backward(): multithreading enabled here:', torch.autograd.is_multithreading_enabled(), flush=True)
    handles = [n.register_prehook(lambda g, n=n: cur.append((n.name(), n._sequence_nr(), threading.get_ident() == main))) for n in nodes]
    try:
        return orig_backward(self, *a, **k)
    finally:
        for h in handles:
            h.remove()


torch.Tensor.backward = logged_backward
outs = []
for it in range(4):
    G.load_state_dict(g0); D.load_state_dict(d0)
    torch.manual_seed(11)
    L = losses.InpaintingLoss(DEV, G, D, noise_mode='const', style_mixing_prob=0)
    phases = ts.make_phases(G, D, kw, kw, g_reg_interval=4, d_reg_interval=16)
    n0 = len(LOG)
    ts.run_phases(real4, 64, phases, batch_idx=0, loss=L, batch_gpu=4, device=DEV)
    outs.append((torch.cat([p.detach().reshape(-1) for p in D.parameters()]).clone(), LOG[n0:]))
    for ph in phases:
        if ph.sync is not None:
            ph.sync.remove()
for it in range(1, 4):
    a, b = outs[0], outs[it]
    print(f'iteration {it} vs 0: differing D elements {int((a[0] != b[0]).sum())}; backward calls {len(b[1])}')
    for ci, (la, lb) in enumerate(zip(a[1], b[1])):
        na, nb = [x[0] for x in la], [x[0] for x in lb]
        first = next((i for i, (x, y) in enumerate(zip(na, nb)) if x != y), None)
        print(f'   backward call {ci}: {len(la)} / {len(lb)} nodes, same name order: {na == nb}, first difference at {first}')
        if first is not None:
            print('      run 0:', la[max(0, first - 2):first + 4]); print(f'      run {it}:', lb[max(0, first - 2):first + 4])
