"""Which tensor ops of the training step launch the at::native kernels (torch.profiler).  usage: python tools/train_step_ops.py [--fp16]"""
import os, sys
sys.argv = [sys.argv[0]] + [a for a in sys.argv[1:]]
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import runpy, torch
from torch.profiler import profile, ProfilerActivity
ns = runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'train_step_bench.py'), run_name='bench')
g_phase, d_phase = ns['g_phase'], ns['d_phase']
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    g_phase(); d_phase()
    torch.cuda.synchronize()
print(prof.key_averages(group_by_input_shape=True).table(sort_by='cuda_time_total', row_limit=int(os.environ.get('SHG_OPS_ROWS', '40')), max_name_column_width=60, max_shapes_column_width=90))
