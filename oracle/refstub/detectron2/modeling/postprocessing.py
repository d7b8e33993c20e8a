def detector_postprocess(results, output_height, output_width):  # PR's own `inference` path; the fixtures use produce_raw_output
    raise RuntimeError("refstub: detector_postprocess is not part of the probabilistic path")
