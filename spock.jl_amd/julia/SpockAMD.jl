# SpockAMD.jl -- thin `ccall` binding of libspock_amd.so (include/spock_amd.h) behind the reference's own API.
#
# UNEXECUTED: there is no Julia toolchain in the build image or on the GPU box, so this file has never been
# run; it is kept short enough to review by eye.  It adds a GPU_SPOCK model type to spock.jl so that
#     model = build_model(scen_tree, cost, dynamics, rms, constraints, SolverOptions(CP, nothing); gpu = true)
#     solve_model!(model, x0, tol = 1e-3)
#     u = model.solver_state.z[model.solver_state_internal.u_inds[1:nu]]      # examples/server_heat/mpc_simulation.jl:164
# keep working unchanged.  Indices are 1-based on the Julia side, 0-based in the C ABI.
module SpockAMD

import LinearAlgebra as LA

const libspock = get(ENV, "SPOCK_AMD_LIB", "libspock_amd.so")

struct SpockProblem            # spock_problem_t
    N::Cint; d::Cint; nx::Cint; nu::Cint; ny::Cint
    anc::Ptr{Cint}; w::Ptr{Cint}
    A::Ptr{Cdouble}; B::Ptr{Cdouble}
    nQ::Cint; nR::Cint; nQN::Cint
    Q::Ptr{Cdouble}; R::Ptr{Cdouble}; QN::Ptr{Cdouble}
    E::Ptr{Cdouble}; F::Ptr{Cdouble}; b::Ptr{Cdouble}
    n_cones::Cint; cone_kind::Ptr{Cint}; cone_dim::Ptr{Cint}
    x_min::Cdouble; x_max::Cdouble; u_min::Cdouble; u_max::Cdouble
end

struct SpockOptions            # spock_options_t
    algorithm::Cint; device::Cint; batch::Cint; split_stage::Cint
    L_norm::Cdouble; max_iter::Cint; graph_iters::Cint
end

struct SpockResult             # spock_result_t
    status::Cint; iterations::Cint; iter_ref::Cint; reserved::Cint
    xi1::Cdouble; xi2::Cdouble; xi::Cdouble; solve_ms::Cdouble
end

mutable struct GPUSolverState  # what scripts read: model.solver_state.z / .v
    nz::Int; nv::Int; L_norm::Float64
    z::Vector{Float64}; v::Vector{Float64}
end

mutable struct GPU_SPOCK
    handle::Ptr{Cvoid}
    solver_state::GPUSolverState
    solver_state_internal            # x_inds, u_inds, s_inds, tau_inds, y_inds (reuse z_to_x, ... of implicit_l.jl:106-158)
    problem_definition
end

check(rc) = rc == 0 || error(unsafe_string(ccall((:spock_last_error, libspock), Cstring, ())))

cone_kind(c) = c isa MOI.Zeros ? 0 : c isa MOI.Nonnegatives ? 1 : c isa MOI.Nonpositives ? 2 : c isa MOI.Reals ? 3 : 4

"""build_model(...) for the GPU path (src/model.jl:273-297)."""
function build_gpu_model(scen_tree, cost, dynamics, rms, constraints, algorithm::Int; device = 0)
    nx, nu = size(dynamics.B[1]); d = length(dynamics.A); n = scen_tree.n
    flat(ms) = all(m -> m == ms[1], ms) ? (1, vec(Matrix(ms[1]))) : (length(ms), vcat((vec(Matrix(m)) for m in ms)...))
    nQ, Q = flat(cost.Q); nR, R = flat(cost.R); nQN, QN = flat(cost.QN)
    A = vcat((vec(a) for a in dynamics.A)...); B = vcat((vec(b) for b in dynamics.B)...)
    anc = Cint[i == 1 ? -1 : scen_tree.anc_mapping[i] - 1 for i in 1:n]
    w = Cint[i == 1 ? -1 : scen_tree.node_info[i].w - 1 for i in 1:n]
    rm = rms[1]; E = vec(rm.E); F = vec(rm.F); b = copy(rm.b)
    kinds = Cint[cone_kind(c) for c in rm.K.subcones]; dims = Cint[MOI.dimension(c) for c in rm.K.subcones]
    h = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve anc w A B Q R QN E F b kinds dims begin
        p = SpockProblem(scen_tree.N, d, nx, nu, length(b), pointer(anc), pointer(w), pointer(A), pointer(B),
                         nQ, nR, nQN, pointer(Q), pointer(R), pointer(QN), pointer(E), pointer(F), pointer(b),
                         length(kinds), pointer(kinds), pointer(dims),
                         constraints.x_min, constraints.x_max, constraints.u_min, constraints.u_max)
        o = SpockOptions(algorithm, device, 1, 0, NaN, 0, 0)
        check(ccall((:spock_create, libspock), Cint, (Ref{SpockProblem}, Ref{SpockOptions}, Ref{Ptr{Cvoid}}), p, o, h))
    end
    sizes = zeros(Cint, 26)
    check(ccall((:spock_sizes, libspock), Cint, (Ptr{Cvoid}, Ptr{Cint}), h[], sizes))
    nz, nv = Int(sizes[4]), Int(sizes[5])
    model = GPU_SPOCK(h[], GPUSolverState(nz, nv, algorithm == 1 ? 10. : 3.28, zeros(nz), zeros(nv)), nothing, nothing)
    finalizer(m -> ccall((:spock_destroy, libspock), Cint, (Ptr{Cvoid},), m.handle), model)
    return model
end

"""solve_model!(model, x0; tol, verbose, z0, v0, gamma, sigma)  (src/models/cpock.jl:169-189)."""
function solve_model!(model::GPU_SPOCK, x0::AbstractVector{Float64}; tol::Float64 = 1e-3, verbose::Int = 1,
                      z0 = nothing, v0 = nothing, gamma = nothing, sigma = nothing)
    g = (gamma === nothing || sigma === nothing) ? NaN : gamma
    s = (gamma === nothing || sigma === nothing) ? NaN : sigma
    res = Ref{SpockResult}()
    zp = z0 === nothing ? C_NULL : pointer(z0); vp = v0 === nothing ? C_NULL : pointer(v0)
    GC.@preserve z0 v0 check(ccall((:spock_solve, libspock), Cint,
        (Ptr{Cvoid}, Ptr{Cdouble}, Cdouble, Cint, Ptr{Cdouble}, Ptr{Cdouble}, Cdouble, Cdouble, Ref{SpockResult}),
        model.handle, x0, tol, verbose, zp, vp, g, s, res))
    check(ccall((:spock_get_z, libspock), Cint, (Ptr{Cvoid}, Ptr{Cdouble}), model.handle, model.solver_state.z))
    check(ccall((:spock_get_v, libspock), Cint, (Ptr{Cvoid}, Ptr{Cdouble}), model.handle, model.solver_state.v))
    println("Finished in $(res[].iter_ref) iterations.")           # cp.jl:230
    return nothing
end

end # module
