"""Run-time state of the stand-in: kernel-scope depth and the hooks the cross-check script sets."""
depth = 0                 # > 0 while a @ti.func / @ti.kernel body runs ("Taichi scope")
rng = None                # callable() -> float in [0,1): what ti.random() returns
pixels = None             # callable(field) -> iterable of index tuples (or None = all) for struct-for loops
on_index = None           # callable(index_tuple) called before each struct-for iteration
imread = None             # callable(path) -> uint8 array (W,H,3): what ti.tools.imread returns
