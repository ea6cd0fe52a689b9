"""Where the HOST time of a training phase goes (cProfile over the G phase of tools/train_step_bench.py).  usage: python tools/train_host_profile.py [--fp16]"""
import cProfile, os, pstats, runpy, sys
sys.argv = [sys.argv[0]] + sys.argv[1:] + ['--steps', '1']
ns = runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'train_step_bench.py'), run_name='bench')
import torch
g_phase = ns['g_phase']
g_phase(); torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
g_phase()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(35)
st.sort_stats('cumulative').print_stats(45)
