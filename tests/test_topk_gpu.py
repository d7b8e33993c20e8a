"""pod_level_topk through the C ABI against torch.sort: exact top-k of every level's key list, sorted descending, for the
single-workgroup path (<= 2048 candidates), the sliced path (16 workgroups + last-one-merges) and its corner sizes; the
level-concatenated copy (cat_keys / cat_level / n_total) the gather kernels read; tickets left zeroed, counts untouched
(the gather kernel consumes them)."""
import pytest
import torch

from pod_compare_amd import hip

pytestmark = pytest.mark.gpu


def make_keys(C, seed, skew):
    g = torch.Generator().manual_seed(seed)
    if skew:      # scores piled up in a narrow band: many keys share their top bytes (what a radix select finds hardest)
        scores = 0.7 + 1e-4 * torch.rand(C, generator=g)
    else:
        scores = torch.rand(C, generator=g) * 0.95 + 0.05
    idx = torch.randperm(C, generator=g)
    return (scores.view(torch.int32).to(torch.int64) << 32) | (0xFFFFFFFF - idx.to(torch.int64))


@pytest.mark.parametrize("skew", [False, True])
@pytest.mark.parametrize("counts,topk", [([1, 0, 5], 1000), ([63, 64, 65], 1000), ([700, 1024, 1500], 1000), ([2048, 2049, 5000], 1000),
                                         ([20000, 2268, 594], 1000), ([145152, 36288, 9072, 2268, 594], 1000),
                                         # the register-cached select's size classes: one workgroup (<= 16 384), 16 cached slices (<= 262 144),
                                         # the loop version beyond; k > 1024 makes the slices hand over 2048 survivors (final selection: loop version)
                                         ([16384, 16385, 2050], 1000), ([300000, 40000], 1000), ([145152, 9072, 3000], 2048), ([36288, 16000], 100),
                                         ([5000, 20000], 1), ([2049, 40000], 1024), ([2049, 40000], 1025)])
def test_level_topk_equals_sorted_prefix(counts, topk, skew):
    lib, P = hip.load(), hip.ptr
    L = len(counts)
    cfg = hip.PodConfig()
    cfg.n_levels, cfg.topk = L, topk
    lv = (hip.PodLevel * L)()
    base, bases = 0, []
    for l, c in enumerate(counts):
        lv[l].anchor_base = base
        bases.append(base)
        base += c + 3                      # odd gaps: level lists need not be 16-byte aligned
    keys = torch.zeros(base, dtype=torch.int64)
    refs = []
    for l, c in enumerate(counts):
        kk = make_keys(c, 10 * l + c, skew)
        keys[bases[l]:bases[l] + c] = kk
        refs.append(torch.sort(kk, descending=True)[0][:topk])
    dk = keys.cuda()
    cnt = torch.zeros(2 * L, dtype=torch.int32, device="cuda")
    cnt[:L] = torch.tensor(counts, dtype=torch.int32)
    sel = torch.full((L * topk,), -1, dtype=torch.int64, device="cuda")
    sc = torch.full((L,), -1, dtype=torch.int32, device="cuda")
    cat = torch.full((L * topk,), -1, dtype=torch.int64, device="cuda")
    cat_lv = torch.full((L * topk,), -1, dtype=torch.int32, device="cuda")
    nt = torch.full((1,), -1, dtype=torch.int32, device="cuda")
    for rnd in range(2):                   # second round: the gather kernel has consumed the counts, every level comes out empty
        hip.check(lib.pod_level_topk(cfg, lv, P(dk), P(cnt), P(sel), P(sc), P(cat), P(cat_lv), P(nt), hip.current_stream()), "pod_level_topk")
        torch.cuda.synchronize()
        assert int(cnt[L:].abs().sum()) == 0                       # tickets
        if rnd == 0:
            assert cnt[:L].cpu().tolist() == counts                # only read here
            assert sc.cpu().tolist() == [min(topk, c) for c in counts]
            for l in range(L):
                assert torch.equal(sel[l * topk:l * topk + len(refs[l])].cpu(), refs[l]), l
            n = sum(min(topk, c) for c in counts)
            assert int(nt.item()) == n
            assert torch.equal(cat[:n].cpu(), torch.cat(refs))
            assert cat_lv[:n].cpu().tolist() == [l for l, r in enumerate(refs) for _ in range(len(r))]
            cnt[:L] = 0                                            # what pod_gather_candidates / pod_gather_decode do
        else:
            assert sc.cpu().tolist() == [0] * L and int(nt.item()) == 0
