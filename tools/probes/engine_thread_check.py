import threading, torch
class F(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x * 2
    @staticmethod
    def backward(ctx, g):
        print('   backward on main thread:', threading.get_ident() == MAIN, ' multithreading enabled (this thread):', torch.autograd.is_multithreading_enabled())
        return F.apply(g)
MAIN = threading.get_ident()
x = torch.randn(4, device='cuda', requires_grad=True)
for mode in (True, False):
    print('set_multithreading_enabled(%s)' % mode)
    with torch.autograd.set_multithreading_enabled(mode):
        F.apply(x).sum().backward()
        (g,) = torch.autograd.grad((F.apply(x) ** 2).sum(), x, create_graph=True)
        print('   node created in backward: sequence_nr', (g * 1).grad_fn.next_functions[0][0]._sequence_nr(), ' forward node created next:', (x * 1).grad_fn._sequence_nr())
