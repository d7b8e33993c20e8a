"""Host logic of pod_compare_amd.conv1x1 that needs no GPU: the split policy of the 1x1 convolutions (Conv1x1.splits_for) on the shapes of a
ResNet-50-FPN at the benchmark frame -- the column `tools/conv1x1_splits.py` measured as the fastest -- and its invariants on arbitrary shapes."""
import itertools

from pod_compare_amd.conv1x1 import Conv1x1


class Shape(Conv1x1):
    def __init__(self, cin, cout):          # (no weights: the policy reads C and K only)
        self.C, self.K = cin, cout


# (Cin, Cout, output pixels) -> (splits over workgroup sets, wavefronts sharing the K range inside a workgroup): the fastest cell of every row of
# profiles/r05_conv1x1_splits.txt (tools/conv1x1_splits.py on an MI355X, replayed as graphs; ties within 2 % go to fewer workgroup sets)
MEASURED_BEST = [((64, 64, 64512), (1, 1)), ((256, 64, 64512), (1, 1)), ((64, 256, 64512), (1, 1)), ((256, 128, 16128), (1, 2)), ((512, 128, 16128), (1, 2)),
                 ((128, 512, 16128), (1, 1)), ((256, 512, 16128), (1, 1)), ((512, 256, 4032), (1, 4)), ((1024, 256, 4032), (1, 4)), ((256, 1024, 4032), (1, 1)),
                 ((512, 1024, 4032), (1, 1)), ((1024, 512, 1008), (1, 4)), ((2048, 512, 1008), (2, 4)), ((512, 2048, 1008), (1, 2)), ((1024, 2048, 1008), (1, 2)),
                 ((512, 256, 16128), (1, 1)), ((2048, 256, 1008), (4, 4))]


def test_policy_picks_the_measured_optimum_on_the_backbone_shapes():
    for (cin, cout, px), want in MEASURED_BEST:
        s = Shape(cin, cout).splits_for(px)
        tiles = ((px + 63) // 64) * (cout // 64)
        assert (s, Conv1x1.auto_waves(tiles * s, (cin // 16) // s)) == want, (cin, cout, px)


def test_policy_invariants():
    for cin, cout, px in itertools.product((16, 48, 64, 256, 1000 * 16, 2048), (64, 256, 2048), (1, 63, 64, 1000, 70000)):
        s = Shape(cin, cout).splits_for(px)
        nks = cin // 16
        assert s in (1, 2, 4, 8, 16) and nks % s == 0                       # what pod_conv1x1_split accepts
        wv = Conv1x1.auto_waves(((px + 63) // 64) * (cout // 64) * s, nks // s)
        assert wv in (1, 2, 4) and (wv == 1 or ((nks // s) % (2 * wv) == 0 and nks // s // wv >= 4))     # whole pairs of k-steps per wavefront
        if s > 1:
            assert nks // s >= 4                                           # never down to a prologue-only loop
            assert ((px + 63) // 64) * (cout // 64) * s <= 1024            # never past one wavefront per SIMD
