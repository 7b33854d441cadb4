# RANSACHIP.jl -- the `ccall` shim a RANSAC.jl maintainer adds to route the hot path of
# RANSAC.jl v0.6.0 through libransac_hip.so (include/ransac_hip.h).
#
# NOT EXECUTED IN THIS REPOSITORY'S CI: the build image has no Julia runtime.  It is written
# against RANSAC.jl v0.6.0's source (src/fitting.jl, src/iterations.jl, src/shapes/*.jl) and the
# C ABI; struct layouts below must stay in sync with include/ransac_hip.h (tests/test_abi.py pins
# sizeof(rh_shape) == 88 and the rh_params field offsets on the C side).
#
# Usage:
#   using RANSAC, RANSACHIP
#   pc  = RANSACCloud(vs, ns, 32)
#   hpc = RANSACHIP.HIPCloud(pc)                      # uploads once; pc stays the source of truth on the host
#   extracted, secs = RANSACHIP.ransac(hpc, params)   # same loop as RANSAC.ransac, GPU hot path
module RANSACHIP

using RANSAC
using RANSAC: FittedShape, FittedPlane, FittedSphere, FittedCylinder, FittedCone,
              ExtractedShape, IterationCandidates, ConfidenceInterval, E,
              recordscore!, findhighestscore, forcefitshapes!, samplepointcloud4!,
              chooseS, prob, updatelevelweight, strt
using StaticArrays

const LIB = get(ENV, "RANSAC_HIP_LIB", joinpath(@__DIR__, "..", "ransac.jl_amd", "libransac_hip.so"))

const RH_PLANE, RH_SPHERE, RH_CYLINDER, RH_CONE = Cint(0), Cint(1), Cint(2), Cint(3)

# typedef struct { int32_t kind; int32_t outwards; double v[10]; } rh_shape;   (88 bytes)
struct RhShape
    kind::Cint
    outwards::Cint
    v::NTuple{10,Cdouble}
end

# rh_params, field for field (include/ransac_hip.h)
struct RhParams
    eps::NTuple{4,Cdouble}
    alpha::NTuple{4,Cdouble}
    cos_alpha::NTuple{4,Cdouble}
    collin_threshold::Cdouble
    parallelthrdeg::Cdouble
    cos_parallelthr::Cdouble
    sphere_par::Cdouble
    minconeopang::Cdouble
    prob_det::Cdouble
    tau::Int64
    itermax::Int64
    drawN::Cint
    minsubsetN::Cint
    extract_s::Cint
    terminate_s::Cint
    n_shape_types::Cint
    shape_types::NTuple{8,Cint}
    score_mode::Cint
    sphere_uses_enabled::Cint
    sampling_streams::Cint
    octree_sampling::Cint
    octree_max_depth::Cint
end

lasterror() = unsafe_string(ccall((:rh_last_error, LIB), Cstring, ()))
check(rc) = rc == 0 ? nothing : error("libransac_hip error $rc: $(lasterror())")

pad10(xs...) = ntuple(i -> i <= length(xs) ? Cdouble(xs[i]) : 0.0, 10)

# FittedShape -> rh_shape.  cos/sin(-opang/2) are computed HERE, with Julia's libm, exactly as
# rodrigues() would (src/utilities.jl:21-22), so the device sees the reference's own constants.
toC(s::FittedPlane) = RhShape(RH_PLANE, 0, pad10(s.point..., s.normal...))
toC(s::FittedSphere) = RhShape(RH_SPHERE, s.outwards, pad10(s.center..., s.radius))
toC(s::FittedCylinder) = RhShape(RH_CYLINDER, s.outwards, pad10(s.axis..., s.center..., s.radius))
toC(s::FittedCone) = RhShape(RH_CONE, s.outwards,
    pad10(s.apex..., s.axis..., s.opang, cos(-s.opang/2), sin(-s.opang/2)))

kindof(::Type{<:FittedPlane}) = RH_PLANE
kindof(::Type{<:FittedSphere}) = RH_SPHERE
kindof(::Type{<:FittedCylinder}) = RH_CYLINDER
kindof(::Type{<:FittedCone}) = RH_CONE

# nested NamedTuple (src/utilities.jl:332-399) -> rh_params; thresholds use Julia's cos / cosd
function toC(p::NamedTuple; score_mode = 0, sphere_uses_enabled = 0, sampling_streams = 0)
    get2(nt, k, d) = haskey(nt, k) ? getfield(nt, k) : d
    sh(name) = get2(p, name, (ϵ = 0.3, α = deg2rad(5)))
    order = (:plane, :sphere, :cylinder, :cone)                       # RH_* kind order
    eps = ntuple(i -> Cdouble(sh(order[i]).ϵ), 4)
    alp = ntuple(i -> Cdouble(sh(order[i]).α), 4)
    it, co = p.iteration, p.common
    sym = Dict(:lengthC => 1, :allcand => 2, :nofminset => 3)
    st = [kindof(T) for T in it.shape_types]
    RhParams(eps, alp, ntuple(i -> cos(alp[i]), 4),
        co.collin_threshold, co.parallelthrdeg, cosd(co.parallelthrdeg),
        get2(get2(p, :sphere, NamedTuple()), :sphere_par, 0.02),
        get2(get2(p, :cone, NamedTuple()), :minconeopang, deg2rad(2)),
        it.prob_det, it.τ, it.itermax, it.drawN, it.minsubsetN,
        sym[it.extract_s], sym[it.terminate_s], length(st),
        ntuple(i -> i <= length(st) ? st[i] : Cint(0), 8), score_mode, sphere_uses_enabled, sampling_streams, 0, 10)
end

"Device-resident twin of a RANSACCloud; `pc` stays authoritative for fits and sampling."
mutable struct HIPCloud{P}
    pc::P
    handle::Ptr{Cvoid}
    function HIPCloud(pc; device = 0)
        # Vector{SVector{3,Float64}} is n x 3 contiguous doubles: passed as is (zero copy on the host side)
        h = Ref{Ptr{Cvoid}}(C_NULL)
        s1 = pc.subsets[1]
        if eltype(eltype(pc.vertices)) == Float32
            # RANSACCloud(...; force_eltype = Float32): Vector{SVector{3,Float32}} as is; scoring and refit then run in binary32
            GC.@preserve pc check(ccall((:rh_cloud_create_f32, LIB), Cint,
                (Ptr{Cfloat}, Ptr{Cfloat}, Int64, Ptr{Int64}, Int64, Cint, Ptr{Ptr{Cvoid}}),
                pointer(reinterpret(Float32, pc.vertices)), pointer(reinterpret(Float32, pc.normals)),
                pc.size, s1, length(s1), device, h))
        else
            GC.@preserve pc check(ccall((:rh_cloud_create, LIB), Cint,
                (Ptr{Cdouble}, Ptr{Cdouble}, Int64, Ptr{Int64}, Int64, Cint, Ptr{Ptr{Cvoid}}),
                pointer(reinterpret(Float64, pc.vertices)), pointer(reinterpret(Float64, pc.normals)),
                pc.size, s1, length(s1), device, h))
        end
        obj = new{typeof(pc)}(pc, h[])
        push_enabled!(obj)
        finalizer(o -> ccall((:rh_cloud_destroy, LIB), Cint, (Ptr{Cvoid},), o.handle), obj)
    end
end

# pc.isenabled::BitVector -> device (same chunk layout), and back
push_enabled!(h::HIPCloud) = check(ccall((:rh_cloud_set_enabled, LIB), Cint,
    (Ptr{Cvoid}, Ptr{UInt64}, Int64), h.handle, h.pc.isenabled.chunks, length(h.pc.isenabled.chunks)))
pull_enabled!(h::HIPCloud) = check(ccall((:rh_cloud_get_enabled, LIB), Cint,
    (Ptr{Cvoid}, Ptr{UInt64}, Int64), h.handle, h.pc.isenabled.chunks, length(h.pc.isenabled.chunks)))

"""
Replacement for `scorecandidates!` (src/fitting.jl:181-190): ONE batched call instead of a
sequential loop.  Legal because nothing reads a score before the loop ends (iterations.jl:99).
`inpoints` are rebuilt from the returned bit masks (subset order), so `IterationCandidates` and
`removeinvalidshapes!` keep working unchanged.
"""
function scorecandidates!(h::HIPCloud, ic::IterationCandidates, candidates, subsetID, params, octree_levels;
                          cparams = toC(params))
    @assert subsetID == 1 "only subset 1 is resident on the device (iterations.jl:95)"
    pc = h.pc
    b = length(candidates)
    if b > 0
        shapes = RhShape[toC(c) for c in candidates]
        counts = Vector{Int32}(undef, b)
        s1 = pc.subsets[1]
        w = cld(length(s1), 64)
        masks = Matrix{UInt64}(undef, w, b)               # column i = row i of the C array
        check(ccall((:rh_score_batch, LIB), Cint,
            (Ptr{Cvoid}, Ptr{RhShape}, Int32, Ref{RhParams}, Ptr{Int32}, Ptr{UInt64}),
            h.handle, shapes, b, cparams, counts, masks))
        for i in 1:b
            bits = BitVector(undef, length(s1))
            copyto!(bits.chunks, view(masks, :, i))
            ip = s1[bits]
            sc = RANSAC.estimatescore(length(s1), pc.size, length(ip))
            pc.levelscore[octree_levels[i]] += E(sc)
            recordscore!(ic, candidates[i], sc, ip)
        end
    end
    empty!(candidates); empty!(octree_levels)
    return nothing
end

"Replacement for `refit` (src/shapes/*.jl): full-cloud scan on the device, ascending indices."
function refit(s::FittedShape, h::HIPCloud, params; cparams = toC(params))
    idx = Vector{Int}(undef, h.pc.size)
    n = Ref{Int64}(0)
    check(ccall((:rh_refit, LIB), Cint,
        (Ptr{Cvoid}, Ref{RhShape}, Ref{RhParams}, Ptr{Int64}, Int64, Ptr{Int64}),
        h.handle, toC(s), cparams, idx, length(idx), n))
    resize!(idx, n[])
    return ExtractedShape(s, idx)
end

"Replacement for `invalidate_indexes!` (src/fitting.jl:197-202): host bits and device bits."
function invalidate_indexes!(h::HIPCloud, indexlist)
    RANSAC.invalidate_indexes!(h.pc, indexlist)
    check(ccall((:rh_invalidate, LIB), Cint, (Ptr{Cvoid}, Ptr{Int64}, Int64), h.handle, indexlist, length(indexlist)))
end

"""
`ransac(pc, params)` (src/iterations.jl:35-162) with the three hot calls swapped; sampling, `fit`,
`findhighestscore`, `prob`, `removeinvalidshapes!` are the reference's own functions.
"""
function ransac(h::HIPCloud, params; reset_rand = false, batched_sampling = false, seed::Integer = 1234)
    pc = h.pc
    reset_rand && RANSAC.Random.seed!(1234)
    it = params.iteration
    cparams = toC(params)
    push_enabled!(h)
    start_time = time_ns()
    candidates = FittedShape[]; scoredshapes = IterationCandidates(); extracted = ExtractedShape[]
    levels = Int[]; sd = Vector{Int}(undef, it.drawN); countcandidates = [0, 0, 0]
    rng = RhRng((UInt64(0), UInt64(0), UInt64(0), UInt64(0)), C_NULL, 0, 0, 0)
    batched_sampling && ccall((:rh_rng_seed, LIB), Cvoid, (Ref{RhRng}, UInt64), rng, UInt64(seed))
    for k in 1:it.itermax
        count(pc.isenabled) < it.τ && break
        if batched_sampling
            # the iteration's minsubsetN calls of samplepointcloud4! as ONE launch (rh_sample_sets) on the library's generator:
            # the same sets and the same number of draws as minsubsetN sequential calls, without a round trip per point
            sets, ok, lev = sample_sets!(h, rng, it.drawN, it.minsubsetN)
            for i in 1:it.minsubsetN
                ok[i] != 0 || continue
                sdi = view(sets, :, i)
                forcefitshapes!(view(pc.vertices, sdi), view(pc.normals, sdi), params, candidates, levels, Int(lev[i]), pc)
            end
        end
        for i in 1:(batched_sampling ? 0 : it.minsubsetN)
            res = samplepointcloud4!(sd, pc, params)
            res[1] || continue
            forcefitshapes!(view(pc.vertices, sd), view(pc.normals, sd), params, candidates, levels, res[2], pc)
        end
        countcandidates[2] += length(candidates)
        scorecandidates!(h, scoredshapes, candidates, 1, params, levels; cparams = cparams)
        countcandidates[3] = k * it.minsubsetN
        countcandidates[1] = length(scoredshapes)
        if length(scoredshapes) > 0
            best = findhighestscore(scoredshapes)
            bestshape = scoredshapes.shapes[best.index]
            scr = E(scoredshapes.scores[best.index])
            if prob(scr, chooseS(countcandidates, it.extract_s), pc.size, it.drawN) > it.prob_det
                ex = refit(bestshape, h, params; cparams = cparams)
                invalidate_indexes!(h, ex.inpoints)
                push!(extracted, ex)
                deleteat!(scoredshapes, best.index)
                RANSAC.removeinvalidshapes!(pc, scoredshapes)
            end
        end
        updatelevelweight(pc)
        prob(it.τ, chooseS(countcandidates, it.terminate_s), pc.size, it.drawN) > it.prob_det && break
    end
    return extracted, trunc((time_ns() - start_time) / 1_000_000_000, digits = 2)
end

"""
`samplepointcloud4!` (src/fitting.jl:383-430) for the `k` minimal sets of an iteration as ONE launch: `rh_sample_sets`.
Returns (`drawN x k` point indices, accept flags, octree levels); `rng` (an `RhRng`, `rh_rng_seed`) is advanced exactly as
`k` sequential calls would advance it.
"""
function sample_sets!(h::HIPCloud, rng, drawN::Integer, k::Integer)
    idx = Matrix{Int64}(undef, drawN, k); ok = Vector{Int32}(undef, k); lev = Vector{Int32}(undef, k)
    check(ccall((:rh_sample_sets, LIB), Cint, (Ptr{Cvoid}, Int32, Ref{RhRng}, Int32, Ptr{Int64}, Ptr{Int32}, Ptr{Int32}),
                h.handle, drawN, rng, k, idx, ok, lev))
    return idx, ok, lev
end

"""
`rh_set_option`: a tuning option for one cloud (or process-wide with `h = nothing`); the library reads no environment
variable.  Keys: "score_path" (0 auto / 1 brute / 2 groups; process-wide, before the cloud is created), "refit_path"
(0 auto / 1 scan / 2 culled), "s4_rows", "unp_words"; `typemin(Int64)` clears a setting.
"""
set_option(h::Union{HIPCloud,Nothing}, key::AbstractString, value::Integer) =
    check(ccall((:rh_set_option, LIB), Cint, (Ptr{Cvoid}, Cstring, Int64), h === nothing ? C_NULL : h.handle, key, value))

# ---- the whole loop on the device: rh_ransac (driver.hip) -------------------------------------
# Mirrors of rh_rng / rh_extracted / rh_result (include/ransac_hip.h).  The index lists live in one
# pinned block owned by the result; they are copied into Julia vectors here and the block goes back
# to the library's pool with rh_result_free.
mutable struct RhRng
    s::NTuple{4,UInt64}
    stream::Ptr{UInt64}
    stream_len::Int64
    stream_pos::Int64
    draws::Int64
end

struct RhExtracted
    shape::RhShape
    n_inpoints::Int64
    inpoints::Ptr{Int64}
    score_E::Cdouble
    iteration::Int64
end

mutable struct RhResult
    shapes::Ptr{RhExtracted}
    n_shapes::Int64
    iterations::Int64
    candidates_scored::Int64
    scored_left::Int64
    seconds::Cdouble
    seconds_score::Cdouble
    seconds_extract::Cdouble
    seconds_host::Cdouble
    seconds_to_last_extraction::Cdouble
    arena::Ptr{Cvoid}
end

fromC(s::RhShape) =
    s.kind == RH_PLANE    ? FittedPlane(SVector(s.v[1:3]...), SVector(s.v[4:6]...)) :
    s.kind == RH_SPHERE   ? FittedSphere(SVector(s.v[1:3]...), s.v[4], s.outwards != 0) :
    s.kind == RH_CYLINDER ? FittedCylinder(SVector(s.v[1:3]...), SVector(s.v[4:6]...), s.v[7], s.outwards != 0) :
                            FittedCone(SVector(s.v[1:3]...), SVector(s.v[4:6]...), s.v[7], s.outwards != 0)

"""
`ransac_device(h, params; seed, sampling_streams)`: the whole `ransac` loop inside the library
(sampling, fits, scoring, refit, invalidation, candidate liveness on the GPU; `sampling_streams = 1`
draws every minimal set from its own counter-based stream so that whole windows of iterations run
on the device).  Same return value as `ransac`.
`mp`: a handle from `mp_open` -- the loop is then run by all processes of that group together on this one
scene (`rh_ransac_mp`: one process per GPU of a node, every one with the same cloud; the minimal sets of every
iteration are dealt round-robin to the ranks); every rank gets the single-process result.
"""
function ransac_device(h::HIPCloud, params; seed::Integer = 1234, sampling_streams::Integer = 1, mp::Ptr{Cvoid} = C_NULL)
    pc = h.pc
    push_enabled!(h)
    cp = Ref(toC(params; sampling_streams = sampling_streams))
    rng = Ref(RhRng((0, 0, 0, 0), C_NULL, 0, 0, 0))
    ccall((:rh_rng_seed, LIB), Cvoid, (Ptr{RhRng}, UInt64), rng, UInt64(seed))
    res = Ref(RhResult(C_NULL, 0, 0, 0, 0, 0.0, 0.0, 0.0, 0.0, 0.0, C_NULL))
    if eltype(eltype(pc.vertices)) === Float32
        # a Float32 cloud (RANSACCloud(...; force_eltype = Float32), octree.jl:102-109): fits, scoring, liveness and refit in
        # binary32, shapes come back holding Float32 values (all four kinds)
        mp == C_NULL || error("ransac_device: mp is not available on a Float32 cloud")
        GC.@preserve pc check(ccall((:rh_ransac_f32, LIB), Cint,
            (Ptr{Cvoid}, Ptr{Cfloat}, Ptr{Cfloat}, Ptr{RhParams}, Ptr{RhRng}, Ptr{RhResult}),
            h.handle, pointer(reinterpret(Cfloat, pc.vertices)), pointer(reinterpret(Cfloat, pc.normals)), cp, rng, res))
    elseif mp == C_NULL
        GC.@preserve pc check(ccall((:rh_ransac, LIB), Cint,
            (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{RhParams}, Ptr{RhRng}, Ptr{RhResult}),
            h.handle, pointer(reinterpret(Cdouble, pc.vertices)), pointer(reinterpret(Cdouble, pc.normals)), cp, rng, res))
    else
        GC.@preserve pc check(ccall((:rh_ransac_mp, LIB), Cint,
            (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{RhParams}, Ptr{RhRng}, Ptr{Cvoid}, Ptr{RhResult}),
            h.handle, pointer(reinterpret(Cdouble, pc.vertices)), pointer(reinterpret(Cdouble, pc.normals)), cp, rng, mp, res))
    end
    extracted = ExtractedShape[]
    for i in 1:res[].n_shapes
        e = unsafe_load(res[].shapes, i)
        push!(extracted, ExtractedShape(fromC(e.shape), copy(unsafe_wrap(Array, e.inpoints, e.n_inpoints))))
    end
    secs = res[].seconds
    ccall((:rh_result_free, LIB), Cvoid, (Ptr{RhResult},), res)
    pull_enabled!(h)          # the cloud's isenabled now reflects the extractions
    return extracted, secs
end

# the processes of one node that share a scene (collective: every rank calls it with the same name; rank 0 creates
# the shared-memory segment).  `MPI.Comm_rank` / `Comm_size` or the launcher's environment give rank and world.
function mp_open(name::AbstractString, rank::Integer, world::Integer; slot_bytes::Integer = 0)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:rh_mp_open, LIB), Cint, (Cstring, Int32, Int32, Int64, Ptr{Ptr{Cvoid}}), name, rank, world, slot_bytes, h))
    return h[]
end
mp_close(mp::Ptr{Cvoid}) = ccall((:rh_mp_close, LIB), Cint, (Ptr{Cvoid},), mp)

# ---- scoring one batch on all GPUs of a node: candidates are independent (src/fitting.jl:181-190), every rank (one
# Julia process per GPU, each with the same cloud) scores the slice lo:hi of the batch and the library's own RCCL
# all-reduce (sum of the zero-padded Int32 counts over xGMI) gives every rank every count.
#   rank 0:      id = comm_unique_id()      ... send the 128 bytes to the other ranks (MPI.Bcast!, a file, a socket) ...
#   every rank:  comm = comm_create(h, rank, world, id)
#   per batch:   counts = scorecounts_sharded(h, comm, candidates, lo, hi, params)      # length(candidates) counts
function comm_unique_id()
    id = zeros(UInt8, 128)
    check(ccall((:rh_comm_unique_id, LIB), Cint, (Ptr{UInt8},), id))
    return id
end
function comm_create(h::HIPCloud, rank::Integer, world::Integer, id::Vector{UInt8})
    c = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:rh_comm_create, LIB), Cint, (Ptr{Cvoid}, Int32, Int32, Ptr{UInt8}, Ptr{Ptr{Cvoid}}), h.handle, rank, world, id, c))
    return c[]
end
comm_destroy(comm::Ptr{Cvoid}) = ccall((:rh_comm_destroy, LIB), Cint, (Ptr{Cvoid},), comm)

function scorecounts_sharded(h::HIPCloud, comm::Ptr{Cvoid}, candidates::Vector{<:FittedShape}, lo::Integer, hi::Integer, params)
    btotal = length(candidates)
    mine = RhShape[toC(c) for c in candidates[lo:hi]]
    cp = Ref(toC(params))
    dsh = Ref{Ptr{Cvoid}}(C_NULL); dcn = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:rh_dev_alloc, LIB), Cint, (Ptr{Cvoid}, Int64, Ptr{Ptr{Cvoid}}), h.handle, max(1, length(mine)) * sizeof(RhShape), dsh))
    check(ccall((:rh_dev_alloc, LIB), Cint, (Ptr{Cvoid}, Int64, Ptr{Ptr{Cvoid}}), h.handle, 4 * max(1, btotal), dcn))
    counts = zeros(Int32, btotal)
    try
        check(ccall((:rh_dev_upload, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{RhShape}, Int64), h.handle, dsh[], mine, length(mine) * sizeof(RhShape)))
        check(ccall((:rh_score_batch_allreduce_dev, LIB), Cint,
            (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int32, Int32, Int32, Ptr{RhParams}, Ptr{Cvoid}),
            h.handle, comm, dsh[], length(mine), lo - 1, btotal, cp, dcn[]))
        check(ccall((:rh_comm_sync, LIB), Cint, (Ptr{Cvoid},), comm))
        check(ccall((:rh_dev_download, LIB), Cint, (Ptr{Cvoid}, Ptr{Int32}, Ptr{Cvoid}, Int64), h.handle, counts, dcn[], 4 * btotal))
    finally
        ccall((:rh_dev_free, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), h.handle, dsh[])
        ccall((:rh_dev_free, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), h.handle, dcn[])
    end
    return counts
end

end # module
