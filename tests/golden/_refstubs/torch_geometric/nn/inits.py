def zeros(t):  # nn/tangent_nonlin.py:5 imports the name only
    if t is not None:
        t.data.fill_(0)
