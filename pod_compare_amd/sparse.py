"""Sparse bbox tower (csrc/k15_sparse_blocks.hip): the blocks of the bbox subnet / bbox_pred / bbox_cov launches that can reach a candidate.

probabilistic_inference.py:310-331 reads box_delta / box_reg_var only at the candidates of :300-308; the head
(probabilistic_retinanet.py:518-537) evaluates the bbox side densely.  `LiveBlocks` is made right after K2 (pod_level_topk) selected the
image's candidates and answers, per (block table, reach), with the device list of the table's live records -- PodWinoConv.live_blocks.
Everything stays on the device: the grids are sized for the whole tables and the surplus workgroups exit."""
from typing import Dict, Tuple

import torch

from . import hip

# reach of a launch = how many convolution layers its OUTPUT lies below the predictors' output: predictors 0, subnet conv 4 .. conv 1: 1 .. 4
REACH_PREDICTOR = 0
DENSE_INPUT = 255                 # in_reach of a launch whose input is dense (the FPN features in front of a subnet's first convolution)
LIVE_HEAD, LIVE_STRIDE = 4, 12    # include/pod_mi355x.h: POD_SPARSE_LIVE_HEAD / _STRIDE -- {count, -, -, -}, then {record, 11 words of need bits} per live record


def reach_of_subnet_layer(layer: int, num_convs: int = 4) -> int:
    """layer 0 .. num_convs - 1 of a subnet -> reach num_convs - layer."""
    return num_convs - layer


class LiveBlocks:
    """The live-record lists of one image (one hot-path workspace `hp`, after its K2 ran on the current stream)."""

    def __init__(self, hp):
        self.hp = hp
        self.cells = sum(h * w for h, w in hp.shapes)
        self.reach = torch.empty(self.cells, dtype=torch.uint8, device=hp.device)
        scratch = torch.empty_like(self.reach)
        self._scratch = scratch                                       # (kept until the launches that use it have been enqueued AND this object dies)
        lv = hp._levels_t()
        for l, (h, w) in enumerate(hp.shapes):
            lv[l].H, lv[l].W, lv[l].anchor_base = h, w, hp.anchor_base[l]
        self._lv = lv
        hip.check(hp.lib.pod_sparse_reach(hp.cfg, lv, hip.ptr(hp.cat_keys), hip.ptr(hp.cat_level), hip.ptr(hp.n_total), hip.ptr(self.reach),
                                          hip.ptr(scratch), hip.current_stream()), "pod_sparse_reach")
        self._lists: Dict[Tuple[int, int], torch.Tensor] = {}

    def __call__(self, table: torch.Tensor, reach: int, in_reach: int = -1) -> torch.Tensor:
        """The device list of `table`'s live records for a launch whose OUTPUT has reach `reach`; cells of its INPUT with a reach above
        `in_reach` are read as 0.0 (default reach + 1: what the layer below computed for this image; DENSE_INPUT: all of it)."""
        in_reach = int(reach) + 1 if in_reach < 0 else int(in_reach)
        key = (table.data_ptr(), int(reach), in_reach)
        live = self._lists.get(key)
        if live is None:
            n = int(table.shape[0])
            live = torch.empty(LIVE_HEAD + LIVE_STRIDE * n, dtype=torch.int32, device=table.device)
            hip.check(self.hp.lib.pod_sparse_live_blocks(self.hp.cfg, self._lv, table.data_ptr(), table.pod_rec_level.data_ptr(), n, hip.ptr(self.reach),
                                                         int(reach), in_reach, hip.ptr(live), hip.current_stream()), "pod_sparse_live_blocks")
            self._lists[key] = live
        return live

    def records(self, table: torch.Tensor, reach: int, in_reach: int = -1) -> torch.Tensor:
        """The live record indices (host sync: tests / diagnostics)."""
        lst = self(table, reach, in_reach)
        return lst[LIVE_HEAD:LIVE_HEAD + LIVE_STRIDE * int(lst[0].item())].view(-1, LIVE_STRIDE)[:, 0]

    def fraction(self, table: torch.Tensor, reach: int) -> float:
        """Live share of a table (host sync: diagnostics only)."""
        return float(self(table, reach)[0].item()) / max(1, int(table.shape[0]))
