# SpockAMD.jl -- `ccall` binding of libspock_amd.so (include/spock_amd.h) behind the reference's own API.
#
# UNEXECUTED: there is no Julia toolchain in the build image or on the GPU box, so this file has never been run; it is kept
# short enough to review by eye against include/spock_amd.h.  It is an INCLUDE file, not a module: add
#       include("models/gpu_spock.jl")          # = this file
# to src/spock.jl after `include("models/spock.jl")` (src/spock.jl:28), so that it can add METHODS to the package's own
# `build_model` / `solve_model!` and subtype its abstract types (src/model.jl:5-21).  A script then changes one line:
#       model = spock.build_model(scen_tree, cost, dynamics, rms, constraints, spock.SolverOptions(spock.SP, nothing), spock.GPU())
#       spock.solve_model!(model, x0, tol = 1e-3)                                            # mpc_simulation.jl:48
#       u = model.solver_state.z[model.solver_state_internal.u_inds[1:nu]]                    # mpc_simulation.jl:164
# Indices are 1-based on the Julia side, 0-based in the C ABI; matrices are column-major on both.

const libspock_amd = get(ENV, "SPOCK_AMD_LIB", "libspock_amd.so")

"""Selects the MI355X path in `build_model` (seventh positional argument)."""
struct GPU
    device::Int      # HIP device ordinal
    batch::Int       # independent instances held by the model (same problem data, own x0 and state)
    L_norm::Float64  # NaN: the reference's hard-coded 10.0 (cpock.jl:63) / 3.28 (models/spock.jl:64); < 0: closed form of ||L||^2
    per_instance_dynamics::Bool   # every instance gets its own A_w, B_w through update_dynamics_batch! (Monte-Carlo draws)
end
GPU(; device = 0, batch = 1, L_norm = NaN, per_instance_dynamics = false) = GPU(device, batch, L_norm, per_instance_dynamics)

struct SpockProblemC           # spock_problem_t (include/spock_amd.h)
    N::Cint; d::Cint; nx::Cint; nu::Cint; ny::Cint
    anc::Ptr{Cint}; w::Ptr{Cint}
    A::Ptr{Cdouble}; B::Ptr{Cdouble}
    nQ::Cint; nR::Cint; nQN::Cint
    Q::Ptr{Cdouble}; R::Ptr{Cdouble}; QN::Ptr{Cdouble}
    E::Ptr{Cdouble}; F::Ptr{Cdouble}; b::Ptr{Cdouble}
    n_cones::Cint; cone_kind::Ptr{Cint}; cone_dim::Ptr{Cint}
    x_min::Cdouble; x_max::Cdouble; u_min::Cdouble; u_max::Cdouble
    n_rms::Cint                # 1: E, F, b hold one risk measure; n_nonleaf: one (ny x d), (ny x d), (ny) block per non-leaf node
end

struct SpockOptionsC           # spock_options_t
    algorithm::Cint; device::Cint; batch::Cint; split_stage::Cint
    L_norm::Cdouble; max_iter::Cint; graph_iters::Cint; batch_layout::Cint
    sp_variant::Cint; sp_max_backtrack::Cint; per_instance_dynamics::Cint
end

struct SpockResultC            # spock_result_t
    status::Cint; iterations::Cint; iter_ref::Cint; reserved::Cint
    xi1::Cdouble; xi2::Cdouble; xi::Cdouble; solve_ms::Cdouble
end

struct SpockSizesC             # spock_sizes_t: 28 x int32
    n::Cint; n_nonleaf::Cint; n_leaf::Cint; nz::Cint; nv::Cint; batch::Cint
    z_x::Cint; z_u::Cint; z_s::Cint; z_tau::Cint; z_y::Cint
    v1::Cint; v2::Cint; v3::Cint; v4::Cint; v5::Cint; v6::Cint; v7::Cint; v11::Cint; v12::Cint; v13::Cint; v14::Cint
    split_stage::Cint; n_subtrees::Cint; graph_iters::Cint; fused_iteration::Cint; interleaved::Cint; layout_reason::Cint
end

# what scripts read: model.solver_state.z / .v (mpc_simulation.jl:50-54); batch > 1: [nz, batch] column per instance
mutable struct GPU_SOLVER_STATE <: SOLVER_STATE
    nz::Int; nv::Int; L_norm::Float64
    z::Array{Float64}; v::Array{Float64}
end

# the index vectors of CP_STATE_INTERNAL / SP_STATE_INTERNAL that scripts use (src/model.jl:80-125), 1-based
struct GPU_STATE_INTERNAL <: SOLVER_STATE_INTERNAL
    x_inds::Vector{Int}; u_inds::Vector{Int}; s_inds::Vector{Int}; tau_inds::Vector{Int}; y_inds::Vector{Int}
end

mutable struct GPU_SPOCK <: CUSTOM_SOLVER_MODEL
    handle::Ptr{Cvoid}
    algorithm::AlgorithmOptions
    batch::Int
    solver_state::GPU_SOLVER_STATE
    solver_state_internal::GPU_STATE_INTERNAL
    problem_definition::GENERIC_PROBLEM_DEFINITION
    results::Vector{SpockResultC}
end

# The C structs above must match include/spock_amd.h field for field (tests/test_abi_layout.py parses both files); the
# library reports its own sizes so that a stale binding fails at load instead of reading past a struct.
function gpu_check_layout()
    sz = zeros(Cint, 4)
    ccall((:spock_abi_struct_sizes, libspock_amd), Cint, (Ptr{Cint}, Cint), sz, 4)
    (sz[1] == sizeof(SpockProblemC) && sz[2] == sizeof(SpockOptionsC) && sz[3] == sizeof(SpockResultC) && sz[4] == sizeof(SpockSizesC)) ||
        error("SpockAMD.jl does not match libspock_amd.so: struct sizes $(sz) vs $(sizeof.((SpockProblemC, SpockOptionsC, SpockResultC, SpockSizesC)))")
end

gpu_check(rc) = rc == 0 || error(unsafe_string(ccall((:spock_last_error, libspock_amd), Cstring, ())))

# ConvexBaseCone kinds (src/risk_measures.jl:6) -> SPOCK_CONE_* (spock_amd.h)
gpu_cone_kind(c) = c isa MOI.Zeros ? 0 : c isa MOI.Nonnegatives ? 1 : c isa MOI.Nonpositives ? 2 : c isa MOI.Reals ? 3 :
                   c isa MOI.SecondOrderCone ? 4 : error("unsupported cone $(typeof(c))")

"""build_model(scen_tree, cost, dynamics, rms, constraints, solver_options, gpu::GPU): the MI355X model (src/model.jl:273-297)."""
function build_model(scen_tree::AbstractScenarioTree, cost::AbstractCost, dynamics::Dynamics,
                     rms::Union{Vector{AbstractRiskMeasure}, Vector{RiskMeasure}}, constraints::AbstractConvexConstraints,
                     solver_options::SolverOptions, gpu::GPU)
    nx, nu = size(dynamics.B[1])
    d = length(dynamics.A)
    n = scen_tree.n
    # `Vector{Matrix}` -> one column-major block per matrix; a vector of identical matrices is passed once (count 1)
    flat(ms) = all(m -> m == ms[1], ms) ? (1, Vector{Float64}(vec(Matrix(ms[1])))) :
                                           (length(ms), Vector{Float64}(vcat((vec(Matrix(m)) for m in ms)...)))
    nQ, Q = flat(cost.Q); nR, R = flat(cost.R); nQN, QN = flat(cost.QN)
    A = Vector{Float64}(vcat((vec(Matrix(a)) for a in dynamics.A)...))
    B = Vector{Float64}(vcat((vec(Matrix(b)) for b in dynamics.B)...))
    anc = Cint[i == 1 ? -1 : scen_tree.anc_mapping[i] - 1 for i in 1:n]        # scenario_tree.jl:90-95, 0-based
    w = Cint[i == 1 ? -1 : scen_tree.node_info[i].w - 1 for i in 1:n]          # scenario_tree.jl:77, 0-based event
    # risk measures: one per non-leaf node (src/risk_measures.jl:93-102 uniform, :127-136 one AV@R per node).  length(b) and the
    # cone K are common to all of them (implicit_l.jl:14,19); identical measures are passed once (n_rms = 1), else stacked
    # node by node: E, F as (ny x d) column-major blocks, b as ny-blocks
    rm = rms[1]
    ny = length(rm.b)
    all(r -> length(r.b) == ny && size(r.E) == size(rm.E) && size(r.F) == size(rm.F), rms) ||
        error("risk measures with differing length(b): not supported on the GPU path")
    kinds = Cint[gpu_cone_kind(c) for c in rm.K.subcones]
    dims = Cint[MOI.dimension(c) for c in rm.K.subcones]
    all(r -> Cint[gpu_cone_kind(c) for c in r.K.subcones] == kinds && Cint[MOI.dimension(c) for c in r.K.subcones] == dims, rms) ||
        error("risk measures with differing cones: not supported on the GPU path")
    uniform_rms = all(r -> r.E == rm.E && r.F == rm.F && r.b == rm.b, rms)
    n_rms = uniform_rms ? 1 : length(rms)                                      # length(rms) == number of non-leaf nodes
    src = uniform_rms ? [rm] : rms
    E = Vector{Float64}(vcat((vec(Matrix{Float64}(r.E)) for r in src)...))
    F = Vector{Float64}(vcat((vec(Matrix{Float64}(r.F)) for r in src)...))
    b = Vector{Float64}(vcat((Vector{Float64}(r.b) for r in src)...))
    h = Ref{Ptr{Cvoid}}(C_NULL)
    gpu_check_layout()
    GC.@preserve anc w A B Q R QN E F b kinds dims begin
        p = SpockProblemC(scen_tree.N, d, nx, nu, ny, pointer(anc), pointer(w), pointer(A), pointer(B),
                          nQ, nR, nQN, pointer(Q), pointer(R), pointer(QN), pointer(E), pointer(F), pointer(b),
                          length(kinds), pointer(kinds), pointer(dims),
                          constraints.x_min, constraints.x_max, constraints.u_min, constraints.u_max, Cint(n_rms))
        o = SpockOptionsC(Int(solver_options.algorithm), gpu.device, gpu.batch, 0, gpu.L_norm, 0, 0, 0, 0, 0, gpu.per_instance_dynamics ? 1 : 0)   # sp_variant 0: as the reference
        gpu_check(ccall((:spock_create, libspock_amd), Cint, (Ref{SpockProblemC}, Ref{SpockOptionsC}, Ref{Ptr{Cvoid}}), p, o, h))
    end
    sz = Ref{SpockSizesC}()
    gpu_check(ccall((:spock_sizes, libspock_amd), Cint, (Ptr{Cvoid}, Ref{SpockSizesC}), h[], sz))
    nz, nv = Int(sz[].nz), Int(sz[].nv)
    Ln = Ref{Cdouble}(0.); Lc = Ref{Cdouble}(0.)
    gpu_check(ccall((:spock_L_norm, libspock_amd), Cint, (Ptr{Cvoid}, Ref{Cdouble}, Ref{Cdouble}), h[], Ln, Lc))
    pd = GENERIC_PROBLEM_DEFINITION(zeros(nx), nx, nu, scen_tree, rms, cost, dynamics, constraints)   # cpock.jl:17-26
    internal = GPU_STATE_INTERNAL(z_to_x(pd), z_to_u(pd), z_to_s(pd), z_to_tau(pd), z_to_y(pd))       # implicit_l.jl:106-158
    @assert last(internal.y_inds) == nz
    shape = gpu.batch == 1 ? (nz,) : (nz, gpu.batch)
    shapev = gpu.batch == 1 ? (nv,) : (nv, gpu.batch)
    model = GPU_SPOCK(h[], solver_options.algorithm, gpu.batch, GPU_SOLVER_STATE(nz, nv, Ln[], zeros(shape), zeros(shapev)),
                      internal, pd, SpockResultC[])
    finalizer(m -> ccall((:spock_destroy, libspock_amd), Cint, (Ptr{Cvoid},), m.handle), model)
    return model
end

function gpu_fetch_state!(model::GPU_SPOCK)
    gpu_check(ccall((:spock_get_z, libspock_amd), Cint, (Ptr{Cvoid}, Ptr{Cdouble}), model.handle, model.solver_state.z))
    gpu_check(ccall((:spock_get_v, libspock_amd), Cint, (Ptr{Cvoid}, Ptr{Cdouble}), model.handle, model.solver_state.v))
end

"""solve_model!(model, x0; tol, verbose, z0, v0, gamma, sigma)  (src/models/cpock.jl:169-189, src/models/spock.jl:231-251).
x0: length nx, or nx x batch.  State persists in the model between calls (warm start), z0 / v0 only together."""
function solve_model!(model::GPU_SPOCK, x0::AbstractArray{TF}; tol::TF = 1e-3, verbose::VERBOSE_LEVEL = SILENT,
                      z0::Union{Array{Float64}, Nothing} = nothing, v0::Union{Array{Float64}, Nothing} = nothing,
                      gamma::Union{Float64, Nothing} = nothing, sigma::Union{Float64, Nothing} = nothing) where {TF <: Real}
    (z0 === nothing) == (v0 === nothing) || error("z0 and v0 must be given together")             # cpock.jl:180-183
    both = gamma !== nothing && sigma !== nothing                                                 # cp.jl:198-201
    x0c = Vector{Float64}(vec(x0))
    length(x0c) == model.problem_definition.nx * model.batch || error("x0 has the wrong length")
    res = Vector{SpockResultC}(undef, model.batch)
    zp = z0 === nothing ? Ptr{Cdouble}(C_NULL) : pointer(z0)
    vp = v0 === nothing ? Ptr{Cdouble}(C_NULL) : pointer(v0)
    GC.@preserve z0 v0 x0c res gpu_check(ccall((:spock_solve, libspock_amd), Cint,
        (Ptr{Cvoid}, Ptr{Cdouble}, Cdouble, Cint, Ptr{Cdouble}, Ptr{Cdouble}, Cdouble, Cdouble, Ptr{SpockResultC}),
        model.handle, x0c, Float64(tol), Cint(Int(verbose)), zp, vp, both ? gamma : NaN, both ? sigma : NaN, res))
    model.results = res
    gpu_fetch_state!(model)                                # model.solver_state.z / .v as the scripts read them
    println("Finished in $(res[1].iter_ref) iterations.")  # cp.jl:230
    return nothing
end

"""The Spock branch of examples/server_heat/mpc_simulation.jl:47-54,164-177 as one call: `size(w_seq, 1)` warm-started solves
per instance; u = z[u_inds[1:nu]] is applied and x0 <- A_w x0 + B_w u advanced on the GPU.  w_seq: steps x batch, 1-based events.
Returns (x_traj [nx, steps + 1, batch], u_traj [nu, steps, batch])."""
function mpc_simulate!(model::GPU_SPOCK, x0::AbstractArray{Float64}, w_seq::AbstractMatrix{<:Integer}; tol::Float64 = 1e-3,
                       verbose::VERBOSE_LEVEL = SILENT)
    steps = size(w_seq, 1)
    nx, nu = model.problem_definition.nx, model.problem_definition.nu
    wc = Matrix{Cint}(w_seq .- 1)                          # [steps, batch] column-major == C's [batch][steps]
    xt = zeros(nx, steps + 1, model.batch); ut = zeros(nu, steps, model.batch)
    res = Vector{SpockResultC}(undef, model.batch * steps)
    x0c = Vector{Float64}(vec(x0))
    GC.@preserve x0c wc xt ut res gpu_check(ccall((:spock_mpc_simulate, libspock_amd), Cint,
        (Ptr{Cvoid}, Ptr{Cdouble}, Cint, Ptr{Cint}, Cdouble, Cint, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{SpockResultC}),
        model.handle, x0c, steps, wc, tol, Cint(Int(verbose)), xt, ut, res))
    model.results = res
    gpu_fetch_state!(model)
    return xt, ut
end

"""Overwrite the persistent state (z, v) of every instance; for SP this also resets what the reference carries between calls."""
function set_state!(model::GPU_SPOCK, z::Array{Float64}, v::Array{Float64})
    GC.@preserve z v gpu_check(ccall((:spock_set_state, libspock_amd), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}), model.handle, z, v))
    gpu_fetch_state!(model)
end

"""Back to the freshly built model (z = v = 0, every carried buffer at its initial value)."""
function reset!(model::GPU_SPOCK)
    gpu_check(ccall((:spock_reset, libspock_amd), Cint, (Ptr{Cvoid},), model.handle))
    gpu_fetch_state!(model)
end

"""New dynamics of the same shape (e.g. a model re-linearised between the steps of a closed loop): ricatti_offline
(implicit_l.jl:46-97) and the sweep matrices are recomputed on the GPU in the model's matrix pool (spock_update_dynamics); costs,
risk measures, constraints and the solver state are kept.  `dynamics.A[w]`, `dynamics.B[w]` as in src/dynamics.jl:1-4."""
function update_dynamics!(model::GPU_SPOCK, dynamics::Dynamics)
    A = Vector{Float64}(vcat((vec(Matrix{Float64}(a)) for a in dynamics.A)...))     # [d][nx*nx], column-major per event
    B = Vector{Float64}(vcat((vec(Matrix{Float64}(b)) for b in dynamics.B)...))     # [d][nx*nu]
    GC.@preserve A B gpu_check(ccall((:spock_update_dynamics, libspock_amd), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}), model.handle, A, B))
    return nothing
end

"""One draw of the dynamics per instance of the batch (`dynamics[b].A[w]`, `.B[w]`): the builds run side by side on the GPU
(spock_update_dynamics_batch); the model must have been built with `GPU(per_instance_dynamics = true)`.  The reference builds one
model per draw (src/model.jl:273-297)."""
function update_dynamics_batch!(model::GPU_SPOCK, dynamics::Vector{<:Dynamics})
    length(dynamics) == model.batch || error("one Dynamics per instance expected")
    A = Vector{Float64}(vcat((vec(Matrix{Float64}(a)) for dy in dynamics for a in dy.A)...))   # [batch][d][nx*nx]
    B = Vector{Float64}(vcat((vec(Matrix{Float64}(b)) for dy in dynamics for b in dy.B)...))   # [batch][d][nx*nu]
    status = zeros(Cint, model.batch)
    GC.@preserve A B status gpu_check(ccall((:spock_update_dynamics_batch, libspock_amd), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cint}), model.handle, A, B, status))
    return nothing
end
