"""The 1x1 convolutions of a ResNet-50-FPN at the benchmark frame (768 x 1344), with how often a forward runs each."""
# (name, Cin, Cout, H_in, W_in, stride, residual, calls per image)
SHAPES = [("res2 conv1 (first)", 64, 64, 192, 336, 1, False, 1), ("res2 conv1", 256, 64, 192, 336, 1, False, 2), ("res2 conv3", 64, 256, 192, 336, 1, True, 3),
          ("res2 shortcut", 64, 256, 192, 336, 1, False, 1), ("res3 conv1 s2", 256, 128, 192, 336, 2, False, 1), ("res3 conv1", 512, 128, 96, 168, 1, False, 3),
          ("res3 conv3", 128, 512, 96, 168, 1, True, 4), ("res3 shortcut s2", 256, 512, 192, 336, 2, False, 1), ("res4 conv1 s2", 512, 256, 96, 168, 2, False, 1),
          ("res4 conv1", 1024, 256, 48, 84, 1, False, 5), ("res4 conv3", 256, 1024, 48, 84, 1, True, 6), ("res4 shortcut s2", 512, 1024, 96, 168, 2, False, 1),
          ("res5 conv1 s2", 1024, 512, 48, 84, 2, False, 1), ("res5 conv1", 2048, 512, 24, 42, 1, False, 2), ("res5 conv3", 512, 2048, 24, 42, 1, True, 3),
          ("res5 shortcut s2", 1024, 2048, 48, 84, 2, False, 1), ("fpn lateral3", 512, 256, 96, 168, 1, False, 1), ("fpn lateral4", 1024, 256, 48, 84, 1, False, 1),
          ("fpn lateral5", 2048, 256, 24, 42, 1, False, 1)]
