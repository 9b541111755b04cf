class Lane:
    pass
