class _Registry:
    """detectron2's Registry as a decorator: remembers the class, returns it unchanged."""

    def __init__(self):
        self.classes = {}

    def register(self, obj=None):
        if obj is None:
            def deco(cls):
                self.classes[cls.__name__] = cls
                return cls
            return deco
        self.classes[obj.__name__] = obj
        return obj

    def get(self, name):
        return self.classes[name]


META_ARCH_REGISTRY = _Registry()
