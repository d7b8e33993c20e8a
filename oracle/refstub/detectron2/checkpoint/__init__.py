class DetectionCheckpointer:
    def __init__(self, model, save_dir=""):
        self.model = model

    def resume_or_load(self, path, resume=True):
        return {}
