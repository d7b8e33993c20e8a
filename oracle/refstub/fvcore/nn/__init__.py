def sigmoid_focal_loss_jit(*a, **k):
    raise RuntimeError("refstub: training losses are out of scope")


def smooth_l1_loss(*a, **k):
    raise RuntimeError("refstub: training losses are out of scope")
