import os, sys
sys.path.insert(0, '/root/repo')
from types import SimpleNamespace
import torch
from nerfool_amd import ops
from nerfool_amd.gnt.transformer_network import GNT
dev='cuda'
R,S,V,depth=64,64,10,8
torch.manual_seed(0)
net = GNT(SimpleNamespace(netwidth=64, trans_depth=depth), in_feat_ch=32, posenc_dim=63, viewenc_dim=63)
blob = ops.pack_gnt_blob(net.state_dict(), depth, dev); mblob = ops.pack_gnt_mfma_blob(blob, depth)
gen = torch.Generator().manual_seed(1)
rgb_feat = torch.randn(R,S,V,35,generator=gen).to(dev); rd = torch.randn(R,S,V,4,generator=gen).to(dev)
mask = (torch.rand(R,S,V,generator=gen)>0.1).float().to(dev); pts=torch.randn(R,S,3,generator=gen).to(dev); ray_d=torch.randn(R,3,generator=gen).to(dev)
a, wa = ops.gnt_fwd(blob, rgb_feat, rd, mask, pts, ray_d, depth, True)
b, wb = ops.gnt_fwd_mfma(mblob, rgb_feat, rd, mask, pts, ray_d, depth, True)
RW_BASE, RW_LAYER, SW_BASE, SW_LAYER = 392, 136, 1288, 1104
rowf = RW_BASE + depth*RW_LAYER; smpf = SW_BASE + depth*SW_LAYER
per_ray = S*V*rowf + S*smpf
wa = wa.view(R, per_ray); wb = wb.view(R, per_ray)
ra = wa[:, :S*V*rowf].view(R, rowf, V, S); rb = wb[:, :S*V*rowf].view(R, rowf, V, S)
sa = wa[:, S*V*rowf:].view(R, smpf, S); sb = wb[:, S*V*rowf:].view(R, smpf, S)
def cmp(name, x, y):
    d=(x-y).abs().max().item(); m=x.abs().max().item()
    print('%-12s maxdiff %.3e  max %.3e  rel %.2e' % (name, d, m, d/max(m,1e-30)))
cmp('RW_R1', ra[:,0:64], rb[:,0:64])
for i in (0, 3, 7):
    lr = RW_BASE + i*RW_LAYER; ls = SW_BASE + i*SW_LAYER
    cmp('L%d VP'%i, ra[:,lr:lr+64], rb[:,lr:lr+64]); cmp('L%d H'%i, ra[:,lr+64:lr+72], rb[:,lr+64:lr+72]); cmp('L%d PROB'%i, ra[:,lr+72:lr+136], rb[:,lr+72:lr+136])
    for nm,o,n in (('XH1',0,64),('RSTD1',64,1),('XH2',65,64),('F',130,256),('G',386,64),('RXH1',450,64),('QH',515,64),('KH',579,64),('VH',643,64),('ML',707,8),('OUTA',715,64),('RXH2',779,64),('F2',844,256)):
        if nm=='G' and i%2==1: continue
        cmp('L%d %s'%(i,nm), sa[:,ls+o:ls+o+n], sb[:,ls+o:ls+o+n])
cmp('AMAX', sa[:,0:64], sb[:,0:64]); cmp('XHF', sa[:,640:704], sb[:,640:704])
d_rgb = torch.randn(R,3,generator=gen).to(dev)
ga = ops.gnt_bwd(blob, rd, mask, d_rgb, wa.reshape(-1), (R,S,V), depth); gb = ops.gnt_bwd(blob, rd, mask, d_rgb, wb.reshape(-1), (R,S,V), depth)
print('grad rel', ((ga-gb).abs().max()/ga.abs().max()).item(), 'rel-L2', ((ga-gb).norm()/ga.norm()).item())
