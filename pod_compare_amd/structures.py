"""`Boxes` / `Instances` containers with detectron2's field names and indexing behaviour.

The reference returns detectron2 `Instances` (probabilistic_inference.py:604-636,
inference_utils.py:39-54).  When detectron2 is importable its classes are used as-is; otherwise these
duck types stand in (same attributes: `image_size`, `pred_boxes.tensor`, `scores`, `pred_classes`,
`pred_cls_probs`, `pred_boxes_covariance`; `has/get/get_fields/to/__len__/__getitem__`).
"""
import torch

try:  # pragma: no cover - detectron2 is not installed in the build image
    from detectron2.structures import Boxes, Instances  # type: ignore
    HAVE_DETECTRON2 = True
except Exception:  # noqa: BLE001
    HAVE_DETECTRON2 = False

    class Boxes:
        def __init__(self, tensor):
            if not isinstance(tensor, torch.Tensor):
                tensor = torch.as_tensor(tensor, dtype=torch.float32)
            tensor = tensor.to(torch.float32)
            if tensor.numel() == 0:
                tensor = tensor.reshape((-1, 4))
            assert tensor.dim() == 2 and tensor.size(-1) == 4, tensor.size()
            self.tensor = tensor

        def to(self, device):
            return Boxes(self.tensor.to(device=device))

        def clone(self):
            return Boxes(self.tensor.clone())

        def area(self):
            b = self.tensor
            return (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])

        def __getitem__(self, item):
            if isinstance(item, int):
                return Boxes(self.tensor[item].view(1, -1))
            return Boxes(self.tensor[item])

        def __len__(self):
            return self.tensor.shape[0]

        @property
        def device(self):
            return self.tensor.device

        def __repr__(self):
            return "Boxes(" + str(self.tensor) + ")"

    class Instances:
        def __init__(self, image_size, **kwargs):
            self._image_size = image_size
            self._fields = {}
            for k, v in kwargs.items():
                self.set(k, v)

        @property
        def image_size(self):
            return self._image_size

        def __setattr__(self, name, val):
            if name.startswith("_"):
                super().__setattr__(name, val)
            else:
                self.set(name, val)

        def __getattr__(self, name):
            if name == "_fields" or name not in self._fields:
                raise AttributeError("Cannot find field '{}' in the given Instances!".format(name))
            return self._fields[name]

        def set(self, name, value):
            if len(self._fields):
                assert len(self) == len(value), "Adding a field of length {} to a Instances of length {}".format(len(value), len(self))
            self._fields[name] = value

        def has(self, name):
            return name in self._fields

        def get(self, name):
            return self._fields[name]

        def get_fields(self):
            return self._fields

        def to(self, *args, **kwargs):
            ret = Instances(self._image_size)
            for k, v in self._fields.items():
                ret.set(k, v.to(*args, **kwargs) if hasattr(v, "to") else v)
            return ret

        def __getitem__(self, item):
            if type(item) == int:
                if item >= len(self) or item < -len(self):
                    raise IndexError("Instances index out of range!")
                item = slice(item, None, len(self))
            ret = Instances(self._image_size)
            for k, v in self._fields.items():
                ret.set(k, v[item])
            return ret

        def __len__(self):
            for v in self._fields.values():
                return len(v)
            raise NotImplementedError("Empty Instances does not support __len__!")
