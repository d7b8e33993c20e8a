"""Stand-in for fvcore (absent here): only the two loss names probabilistic_retinanet.py imports; training is out of scope."""
