from .solver import Options, Problem, Solver, DefaultConvergenceCriteria  # noqa: F401
