import torch, torch.nn.functional as F
dev='cuda'
torch.manual_seed(0)
x = torch.randn(4,3,762,1014, device=dev)          # already reflect-padded input
w = torch.randn(64,3,7,7, device=dev)*0.05
aten = torch.ops.aten
def timed(fn, n=20):
    for _ in range(3): fn()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): out=fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/n*1e3, out
t_f, y = timed(lambda: aten.convolution(x, w, None, [2,2],[0,0],[1,1],False,[0,0],1))
g = torch.randn_like(y)
t_b, dx = timed(lambda: aten.convolution_backward(g, x, w, None, [2,2],[0,0],[1,1],False,[0,0],1,[True,False,False])[0])
# polyphase: 8x8 zero-padded kernel, space-to-depth 2
w8 = F.pad(w, (0,1,0,1))
w2 = w8.reshape(64,3,4,2,4,2).permute(0,1,3,5,2,4).reshape(64,12,4,4).contiguous()
t_s, x2 = timed(lambda: F.pixel_unshuffle(x, 2))
t_f2, y2 = timed(lambda: aten.convolution(x2, w2, None, [1,1],[0,0],[1,1],False,[0,0],1))
t_b2, dx2 = timed(lambda: aten.convolution_backward(g, x2, w2, None, [1,1],[0,0],[1,1],False,[0,0],1,[True,False,False])[0])
t_u, dxs = timed(lambda: F.pixel_shuffle(dx2, 2))
print('direct: fwd %.1f us bwd %.1f us' % (t_f, t_b))
print('polyphase: unshuffle %.1f fwd %.1f bwd %.1f shuffle %.1f us' % (t_s, t_f2, t_b2, t_u))
print('fwd diff', float((y-y2).abs().max()), float(y.abs().max()), 'bwd diff', float((dx-dxs).abs().max()), float(dx.abs().max()))
