def prep(g):
    return g
