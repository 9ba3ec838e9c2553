"""GNTModel container (gnt/model.py:17-123 attribute surface): net_coarse (GNT), net_fine (None with single_net),
feature_net (ResUNet, single_net => one 32-channel map returned twice), switch_to_eval, start_step, checkpoint loading."""
import os

import torch

from ..ibrnet.feature_network import ResUNet
from .transformer_network import GNT


class GNTModel(object):
    def __init__(self, args, load_opt=False, load_scheduler=False, device=None):
        self.args = args
        if device is None:
            device = torch.device('cuda:%d' % getattr(args, 'local_rank', 0))
        self.device = torch.device(device)
        single_net = bool(getattr(args, 'single_net', True))
        self.net_coarse = GNT(args, in_feat_ch=getattr(args, 'coarse_feat_dim', 32), posenc_dim=63, viewenc_dim=63,
                              ret_alpha=getattr(args, 'ret_alpha', False)).to(self.device)
        # single_net: one network serves the coarse and the fine pass (gnt/model.py:29-39)
        self.net_fine = None if single_net else GNT(args, in_feat_ch=getattr(args, 'fine_feat_dim', 32), posenc_dim=63,
                                                    viewenc_dim=63, ret_alpha=True).to(self.device)
        self.feature_net = ResUNet(coarse_out_ch=getattr(args, 'coarse_feat_dim', 32), fine_out_ch=getattr(args, 'fine_feat_dim', 32),
                                   single_net=single_net).to(self.device)
        for net in (self.net_coarse, self.net_fine, self.feature_net):
            if net is None:
                continue
            for p in net.parameters():
                p.requires_grad_(False)
        self.start_step = 0
        ckpt = getattr(args, 'ckpt_path', None)
        if ckpt and os.path.isfile(ckpt) and not getattr(args, 'no_reload', False):
            to_load = torch.load(ckpt, map_location=self.device)
            self.net_coarse.load_state_dict(to_load['net_coarse'])
            if self.net_fine is not None and 'net_fine' in to_load:
                self.net_fine.load_state_dict(to_load['net_fine'])
            self.feature_net.load_state_dict(to_load['feature_net'])
            try:
                self.start_step = int(ckpt[-10:-4])
            except ValueError:
                self.start_step = 0

    def switch_to_eval(self):
        self.net_coarse.eval()
        if self.net_fine is not None:
            self.net_fine.eval()
        self.feature_net.eval()

    def switch_to_train(self):
        self.net_coarse.train()
        self.feature_net.train()
