"""spock.jl_amd -- MI355X-native Chambolle-Pock / SuperMann (Spock) hot path.

Host-side mirror of the reference's user-level API (build_model / solve_model! / warm start) on top of
the C ABI in include/spock_amd.h (libspock_amd.so: hand-written HIP kernels for gfx950).
"""
from .problem import (CONE_NONNEG, CONE_NONPOS, CONE_REALS, CONE_SOC, CONE_ZEROS, ConvexCone, Cost, Dynamics,
                      FlatProblem, RiskMeasure, ScenarioTree, UniformRectangle, flatten_problem,
                      generate_scenario_tree_uniform_branching_factor_v2, get_nonuniform_rms_avar_v2, get_uniform_rms_avar_v2,
                      get_uniform_rms_tv_v2, get_uniform_rms_v2)
from .model import (CP, LOG, SILENT, SP, AlgorithmOptions, Model, QNewtonOptions, SolverOptions, VERBOSE_LEVEL,
                    build_model, mpc_simulation, solve_model, solve_model_bang)
from ._lib import SpockError, load_library
from .examples import L_norm_sq, get_car_fixture, get_random_linear_specs, get_server_heat_specs
from .validate import check_solution

__all__ = [n for n in dir() if not n.startswith("_")]
