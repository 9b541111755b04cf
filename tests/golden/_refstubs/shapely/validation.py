def make_valid(g):
    return g
