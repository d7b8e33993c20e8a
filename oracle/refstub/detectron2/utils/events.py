def get_event_storage():  # training-time logging only (PR:126-166); never reached by the inference fixtures
    raise RuntimeError("refstub: no event storage outside training")
