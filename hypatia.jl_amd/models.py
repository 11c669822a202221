"""Models.Model{Float64} mirror (/root/reference/src/Models/Models.jl:14-66): plain data."""
import numpy as np


class Model:
    def __init__(self, c, A, b, G, h, cones, obj_offset=0.0):
        self.c = np.array(c, dtype=np.float64)
        self.b = np.array(b, dtype=np.float64)
        self.h = np.array(h, dtype=np.float64)
        self.n, self.p, self.q = self.c.shape[0], self.b.shape[0], self.h.shape[0]
        self.A = np.array(A, dtype=np.float64).reshape(self.p, self.n)
        self.G = np.asarray(G, dtype=np.float64).reshape(self.q, self.n)
        self.obj_offset = float(obj_offset)
        self.cones = list(cones)
        self.cone_idxs = []
        prev = 0
        for cone in self.cones:   # build_cone_idxs, Models.jl:54-64
            d = cone.dimension()
            self.cone_idxs.append(slice(prev, prev + d))
            prev += d
        assert prev == self.q
        self.nu = float(sum(cone.get_nu() for cone in self.cones)) if self.cones else 0.0

    def copy(self):
        return Model(self.c, self.A, self.b, self.G, self.h, self.cones, obj_offset=self.obj_offset)
