class RoadNetwork:
    pass
