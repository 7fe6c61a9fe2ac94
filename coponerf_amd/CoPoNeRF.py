"""Drop-in for the reference's `models/CoPoNeRF.py` on MI355X.

    from coponerf_amd import CoPoNeRF
    model = CoPoNeRF.CoPoNeRF(n_view=2).cuda()            # same call as /root/reference train.py:99, test.py:132
    z, rel_pose, flow = model.get_z(model_input)           # models/CoPoNeRF.py:159-206
    out = model(model_input, z=z, rel_pose=rel_pose, val=True, flow=flow)    # models/CoPoNeRF.py:208-576

Constructor signature, parameter names/shapes (so `load_state_dict(ckpt['model'], strict=False)` works,
train.py:113-116), the input dict, the returned dict (keys, shapes, dtypes, `pixel_val` on the CPU) and the
`.H/.W` side effect follow the reference.  The render path is executed by hand-written gfx950 kernels through
coponerf_amd.render.RenderEngine; there is no PyTorch fallback for it.
"""
from __future__ import annotations

from typing import Dict

import torch
import torch.nn as nn

from . import getz as _getz
from .aux_outputs import aux_outputs
from .render import RenderEngine, _rigid_inverse


class _ResBlock(nn.Module):
    """Parameter container of lightfield.ResnetBlockFC (models/lightfield.py:9-61): x + fc_1(relu(fc_0(relu(x))))."""

    def __init__(self, size: int):
        super().__init__()
        self.fc_0 = nn.Linear(size, size)
        self.fc_1 = nn.Linear(size, size)
        nn.init.constant_(self.fc_0.bias, 0.0)
        nn.init.kaiming_normal_(self.fc_0.weight, a=0, mode="fan_in")
        nn.init.constant_(self.fc_1.bias, 0.0)
        nn.init.zeros_(self.fc_1.weight)


class _LightFieldDecoder(nn.Module):
    """Parameter container of lightfield.ResnetFC(d_in=18, d_latent=832, d_hidden=128, n_blocks=3, d_out=3)
    (models/lightfield.py:64-129; built at models/CoPoNeRF.py:103-104).  Evaluated by cpn_linear_f32."""

    def __init__(self, d_in: int, d_latent: int, d_hidden: int, n_blocks: int = 3, d_out: int = 3):
        super().__init__()
        self.lin_in = nn.Linear(d_in, d_hidden)
        self.lin_out = nn.Linear(d_hidden, d_out)
        self.blocks = nn.ModuleList([_ResBlock(d_hidden) for _ in range(n_blocks)])
        self.lin_z = nn.ModuleList([nn.Linear(d_latent, d_hidden) for _ in range(n_blocks)])
        for lin in (self.lin_in, self.lin_out, *self.lin_z):
            nn.init.constant_(lin.bias, 0.0)
            nn.init.kaiming_normal_(lin.weight, a=0, mode="fan_in")


RENDER_PARAM_PREFIXES = ("query_encode_latent", "query_encode_latent_2", "latent_value", "key_map", "key_map_2",
                         "query_embed", "query_embed_2", "query_repeat_embed", "query_repeat_embed_2",
                         "encode_latent", "phi")


class CoPoNeRF(nn.Module):
    def __init__(self, n_view: int = 1, npoints: int = 64, num_hidden_units_phi: int = 128):
        super().__init__()
        self.n_view = n_view
        self.npoints = npoints if npoints else 64              # models/CoPoNeRF.py:24-27
        self.repeat_attention = True
        latent = 256 * 3 + 64                                   # 832
        hidden = 128
        # ---- get_z stack (models/CoPoNeRF.py:33-67): pose head, UFC aggregation, ResNet-34 encoder
        self.cross_attention = _getz.CrossBlock()
        self.pose_regressor, self.rotation_regressor, self.translation_regressor = _getz.make_pose_heads()
        self.feature_cost_aggregation = _getz.UFC()
        self.encoder = _getz.SpatialEncoder()
        # ---- render-path layers (models/CoPoNeRF.py:69-104); unused upstream layers are kept so that
        #      checkpoints round-trip with identical keys.
        self.conv_map = nn.Conv2d(3, 64, kernel_size=7, stride=1, padding=3)
        self.query_encode_latent = nn.Conv2d(latent + 3, latent, 1)
        self.query_encode_latent_2 = nn.Conv2d(latent, latent // 2, 1)
        self.corr_embed = nn.Conv2d(4096, latent, 1)
        self.latent_dim = latent // 2                           # 416
        self.latent_value = nn.Conv2d(self.latent_dim * n_view, self.latent_dim, 1)
        self.key_map = nn.Conv2d(self.latent_dim * n_view, hidden, 1)
        self.key_map_2 = nn.Conv2d(hidden, hidden, 1)
        self.query_embed = nn.Conv2d(16, hidden, 1)
        self.query_embed_2 = nn.Conv2d(hidden, hidden, 1)
        self.latent_avg_query = nn.Conv2d(9 + 16, hidden, 1)
        self.latent_avg_query_2 = nn.Conv2d(hidden, hidden, 1)
        self.latent_avg_key = nn.Conv2d(self.latent_dim, hidden, 1)
        self.latent_avg_key_2 = nn.Conv2d(hidden, hidden, 1)
        self.query_repeat_embed = nn.Conv2d(16 + 128, hidden, 1)
        self.query_repeat_embed_2 = nn.Conv2d(hidden, hidden, 1)
        self.latent_avg_repeat_query = nn.Conv2d(9 + 16 + 128, hidden, 1)
        self.latent_avg_repeat_query_2 = nn.Conv2d(hidden, hidden, 1)
        self.encode_latent = nn.Conv1d(self.latent_dim, 128, 1)
        self.phi = _LightFieldDecoder(n_view * 9, self.latent_dim * n_view, num_hidden_units_phi)
        self.hidden_dim = hidden
        self.num_hidden_units_phi = num_hidden_units_phi
        self._engine = RenderEngine()
        self.H = self.W = None

    @property
    def _param_epoch(self) -> int:
        """Counts RenderEngine.invalidate() calls (load_state_dict, dist.broadcast_parameters, manual): captured get_z
        graphs (graphs.py) are keyed on it, they hold the weights' values too."""
        return self._engine.epoch

    # ------------------------------------------------------------------------------------------
    def _render_params(self) -> Dict[str, torch.Tensor]:
        # the Parameter OBJECTS are stable across .to()/.cuda()/load_state_dict (those rewrite .data in place), so the
        # walk over the 636 parameters of the module tree is done once, not on every ray chunk of a full-image render
        c = self.__dict__.get("_rp_cache")
        if c is not None:
            rp, owners = c
            # a replaced Parameter (`model.key_map.weight = nn.Parameter(...)`, a swapped sub-module) shows up as a
            # different object in its owner's _parameters: ~40 identity comparisons per call
            if all(mod._parameters.get(leaf) is p for (mod, leaf), p in zip(owners, rp.values())):
                return rp
        rp = {k: v for k, v in self.named_parameters() if k.split(".")[0] in RENDER_PARAM_PREFIXES}
        owners = []
        for k in rp:
            path, leaf = k.rsplit(".", 1)
            owners.append((self.get_submodule(path), leaf))
        self.__dict__["_rp_cache"] = (rp, owners)
        return rp

    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        out = super().load_state_dict(state_dict, strict=strict, assign=assign)
        self._engine.invalidate()      # packed fp16 weights / tables are derived from the parameters; bumps _param_epoch
        return out

    def get_z(self, input, val: bool = False, ops=None):
        """Features, estimated relative pose and flows (models/CoPoNeRF.py:159-206):
        ([(2B,256,16,16),(2B,256,32,32),(2B,256,64,64),(2B,64,256,256)], (B,4,4), 4 x (B,2,64,64)).
        The 4-D operators of UFC run on the HIP kernels (coponerf_amd.ufc_ops.HipOps); `ops` is a test hook."""
        if ops is None:
            from .ufc_ops import HipOps
            ops = HipOps()
        return _getz.get_z(self, input, ops)

    def prepare_next(self, input, z, rel_pose, flow=None) -> None:
        """Optional, inference only: announce the pair whose forward(input, z=z, rel_pose=rel_pose, flow=flow) calls come
        AFTER the next forward() call(s) of the current pair.  Its per-pair preparation (camera copy to the host, feature
        tables, flow products) then runs beside the current pair's kernels on a stream of its own, and the first forward()
        on it starts without a device synchronisation (RenderEngine.prepare_next).  Outputs are unchanged; a pair that was
        never announced is prepared inside its first forward() as before."""
        ctx, qry = input["context"], input["query"]
        self._engine.prepare_next(self._render_params(), ctx["cam2world"], ctx["intrinsics"], qry["cam2world"],
                                  qry["intrinsics"], z, rel_pose, flow=flow, width=ctx["rgb"].shape[-2])

    def forward(self, input, z=None, rel_pose=None, val: bool = False, flow=None, debug: bool = False):
        if self.n_view != 2:
            raise NotImplementedError(f"the HIP render path is specialised for n_view=2 (got {self.n_view})")
        if z is None:
            z, rel_pose, flow = self.get_z(input)
        ctx, qry = input["context"], input["query"]
        self.H, self.W = ctx["rgb"].shape[2], ctx["rgb"].shape[3]    # what get_z records (models/CoPoNeRF.py:180)
        rp = self._render_params()
        train = torch.is_grad_enabled() and (any(p.requires_grad for p in rp.values()) or any(t.requires_grad for t in z))
        args = (rp, ctx["cam2world"], ctx["intrinsics"], qry["cam2world"], qry["intrinsics"], qry["uv"], z, rel_pose, val,
                self.npoints, self.H, self.W)
        # same forward kernels either way; the training pass wraps them in autograd Functions
        core = self._engine.render_train(*args) if train else self._engine.render(*args, debug=debug, inp=input, flow=flow)
        out = {"flow": flow, "uv": qry["uv"], "coords": core["coords"]}
        out["pixel_val"] = core["pixel_val_cpu"]                  # models/CoPoNeRF.py:490 (callers expect a CPU tensor)
        out["at_wts"] = [core["at_wt"]]
        host = core["host"]                                       # O(B) inverses done with the host pose algebra
        if train:
            out.update(aux_outputs(input, flow, core["at_wt"], core["pt"], core["Tq"], host["inv_Kq"], host["inv_qc2w"]))
        else:
            out.update(core["aux"])
        out["at_wt"] = core["at_wt"]
        out["valid_mask"] = core["valid_mask"]
        out["rgb"] = core["rgb"]
        out["z"] = z
        flip = host.get("rel_pose_flip")
        # differentiable on the device when a pose loss may need it, else from the cached host pose algebra
        out["rel_pose_flip"] = _rigid_inverse(rel_pose) if (flip is None or (torch.is_grad_enabled() and rel_pose.requires_grad)) else flip
        out["rel_pose"] = rel_pose
        out["gt_rel_pose"] = host["gt_rel_pose"]                  # models/CoPoNeRF.py:570-574
        out["gt_rel_pose_flip"] = host["gt_rel_pose_flip"]
        if debug:
            out["_core"] = core
        return out
