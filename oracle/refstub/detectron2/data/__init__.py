class _Meta:
    thing_dataset_id_to_contiguous_id = {i + 1: i for i in range(7)}
    json_file = ""


class _Catalog:
    def get(self, name):
        return _Meta()


MetadataCatalog = _Catalog()
