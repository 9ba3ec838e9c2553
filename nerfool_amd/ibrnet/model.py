"""Model container with the attribute surface the attack scripts use (ibrnet/model.py:30-123): `.net_coarse`,
`.net_fine`, `.feature_net`, `.switch_to_eval()`, `.start_step`, and checkpoint loading in the reference layout
({'net_coarse', 'net_fine', 'feature_net', ...}; step parsed from the file name).  Device-neutral (the reference
hard-codes cuda:{local_rank}); optimizer / scheduler / DDP wrappers belong to training and are not built."""
import os

import torch

from .feature_network import ResUNet
from .mlp_network import IBRNet


class IBRNetModel(object):
    def __init__(self, args, load_opt=False, load_scheduler=False, device=None):
        self.args = args
        if device is None:
            device = torch.device('cuda:%d' % getattr(args, 'local_rank', 0))
        self.device = torch.device(device)
        feat_c = getattr(args, 'coarse_feat_dim', 32)
        feat_f = getattr(args, 'fine_feat_dim', 32)
        self.net_coarse = IBRNet(args, in_feat_ch=feat_c, n_samples=args.N_samples).to(self.device)
        self.net_fine = None
        if not getattr(args, 'coarse_only', False):
            self.net_fine = IBRNet(args, in_feat_ch=feat_f, n_samples=args.N_samples + args.N_importance).to(self.device)
        self.feature_net = ResUNet(coarse_out_ch=feat_c, fine_out_ch=feat_f,
                                   coarse_only=getattr(args, 'coarse_only', False)).to(self.device)
        self.freeze()
        self.start_step = 0
        ckpt = getattr(args, 'ckpt_path', None)
        if ckpt and os.path.isfile(ckpt) and not getattr(args, 'no_reload', False):
            self.load_model(ckpt)
            try:
                self.start_step = int(ckpt[-10:-4])
            except ValueError:
                self.start_step = 0

    def freeze(self):
        """The attack differentiates w.r.t. the perturbation only: the fused executor runs backward-DATA passes only and
        requires frozen weights (the reference accumulates weight gradients that nobody reads, eval_adv.py:805-810)."""
        for net in (self.net_coarse, self.net_fine, self.feature_net):
            if net is not None:
                for p in net.parameters():
                    p.requires_grad_(False)

    def _nets(self):
        return [n for n in (self.net_coarse, self.net_fine, self.feature_net) if n is not None]

    def switch_to_eval(self):
        for n in self._nets():
            n.eval()

    def switch_to_train(self):
        for n in self._nets():
            n.train()

    def load_model(self, filename, load_opt=False, load_scheduler=False):
        to_load = torch.load(filename, map_location=self.device)
        for name, net in (('net_coarse', self.net_coarse), ('net_fine', self.net_fine)):
            if net is None:
                continue
            missing, unexpected = net.load_state_dict(to_load[name], strict=False)
            assert len(unexpected) == 0, 'unexpected keys: %s' % (unexpected,)
            assert all(k == 'pos_encoding' for k in missing), 'missing keys: %s' % (missing,)
        self.feature_net.load_state_dict(to_load['feature_net'])

    def save_model(self, filename):
        to_save = {'net_coarse': self.net_coarse.state_dict(), 'feature_net': self.feature_net.state_dict()}
        if self.net_fine is not None:
            to_save['net_fine'] = self.net_fine.state_dict()
        torch.save(to_save, filename)
