# Build-owned stand-in package; see models/resnet.py.  Used ONLY by oracle/make_goldens.py.
