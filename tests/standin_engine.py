"""What bench.py's one_pass() needs of a RolloutEngine, without a GPU (every scenario runs its T steps): lets the launch /
dispatch / timing / collection code of bench.py run on a CPU box -- `bench.py --engine-factory tests.standin_engine:make`."""
import numpy as np


class StandInEngine:
    def __init__(self, R, first, E, T):
        self.R, self.first, self.E, self.T = R, first, E, T
        self.passes = 0

    def rollout_async(self, T, do_reset=True):
        assert T == self.T and do_reset
        self.passes += 1

    def synchronize(self):
        pass

    def metrics(self):
        idx = np.arange(self.first, self.first + self.R, dtype=np.float64)
        rows = dict(ego_avg_speed=idx, ego_max_speed=idx * 2, ego_distance_travelled=idx * 3,
                    n_collisions=np.zeros(self.R, np.int32), n_steps=np.full(self.R, self.T, np.int32))
        return rows, None

    def last_launch_stats(self):
        return 2, 1.0

    def last_kernel_ms(self):
        return 1.25

    def close(self):
        pass


def make(R, first, E, T):
    return StandInEngine(R, first, E, T)
