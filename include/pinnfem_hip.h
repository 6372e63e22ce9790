/*
 * pinnfem_hip.h — C ABI of libpinnfem_hip.so (gfx950 / MI355X).
 *
 * The library is the hand-written HIP implementation of ONE path of the reference
 * (rpacheco-blazquez/PINN-FEM): the body of the PINN+GD iteration
 *     assemble_system_torch -> residual/loss -> loss.backward() -> Adam(u), Adam(theta)
 *     -> BC clamp -> monitors/stop test          (FEM/python/fem/solver.py:252-355)
 * The reference has no native code and no FFI; every entry point below names the Python
 * lines it replaces.  A Python binding (ctypes) is in pinn_fem_amd/_capi.py and the stub a
 * reference maintainer would add is shown in INTEGRATION.md.
 *
 * Conventions
 *   - plain C types only; every pointer marked "dev" is a DEVICE pointer owned by the caller
 *     (PyTorch's allocator in our host code); the library allocates nothing and keeps no
 *     state between calls, so every call is re-entrant and graph-capturable;
 *   - `stream` is a hipStream_t passed as void*; calls only enqueue work (no sync);
 *   - return value: 0 = ok, <0 = error (see pf_last_error()); no exceptions cross the ABI;
 *   - all floating point is IEEE float32 unless a field says double.
 */
#ifndef PINNFEM_HIP_H
#define PINNFEM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PF_ABI_VERSION 7

/* error codes */
#define PF_OK 0
#define PF_ERR_ARG (-1)          /* bad argument / inconsistent sizes */
#define PF_ERR_UNSUPPORTED (-2)  /* net shape outside the compiled menu */
#define PF_ERR_HIP (-3)          /* a HIP runtime call failed */

/* dof flag bits (pf_mesh.dof_flags) */
#define PF_DOF_FIXED 1u
#define PF_DOF_MEASURED 2u
#define PF_DOF_SHARED 4u   /* dof of a node that also belongs to another rank's shard (multi-GPU) */
#define PF_DOF_GHOST 8u    /* shared and owned by another rank: excluded from this rank's global sums */

/* MLP engines (pf_problem.wg_mode) */
#define PF_WG_SHUFFLE 0 /* VALU mat-vecs, weight gradients by wave shuffles (slow cross-check) */
#define PF_WG_MFMA 1    /* VALU mat-vecs, weight gradients on v_mfma_f32_16x16x4_f32 (LDS tiles) */
#define PF_WG_MFMA44 2  /* everything on v_mfma_f32_4x4x1_16B_f32, one element per lane (exact f32 fma chains) */
#define PF_WG_MFMA32 3  /* v_mfma_f32_32x32x16_f16 with 2-way split operands (f32-grade products on the f16 matrix
                           cores; default for widths <= PF_N32_WIDTH_MAX) */
#define PF_N32_WIDTH_MAX 30
#define PF_MLP_F32 0
#define PF_MLP_BF16 1

/* element-force formulations */
#define PF_FE_REFERENCE 0 /* 4-term dot per row, the reference's order (nn_assembly.py:96-100) */
#define PF_FE_DELTA 1     /* on d = u_j - u_i: same algebra, no float32 cancellation (opt-in) */

/* Mesh-derived immutable data ("plan").  Replaces the per-iteration Python work of
 * fem/nn_assembly.py:181-205 (dof numbering, centroid, direction cosines) and
 * fem/boundary.py:8-13 (free/fixed partition).  All arrays dev. */
typedef struct pf_mesh {
  int32_t dim;             /* 1 (nn_assembly.py:129-179) or 2 (:181-229) */
  int32_t n_nodes;
  int32_t n_elems;
  int32_t n_dofs;          /* n_nodes*dim */
  const int32_t* conn;     /* [n_elems][2] node ids (i,j) */
  const float* egeo;       /* [n_elems][4] = c2, cs, s2, l0  (float32 of the float64 values the
                              reference computes, nn_assembly.py:64-82; 1-D: 1,0,0,|dx|) */
  const float* ecent;      /* [n_elems][dim] centroid = NN input columns after load_factor */
  const int32_t* adj_ptr;  /* [n_nodes+1] CSR node -> incident elements, ascending element id */
  const int32_t* adj;      /* [adj_ptr[n_nodes]] entries (elem<<1)|end, end=0: node is i */
  const float* f_ext;      /* [n_dofs] external loads (solver.py:218) */
  const uint8_t* dof_flags;/* [n_dofs] PF_DOF_* */
  const float* meas_val;   /* [n_dofs] measured displacement where PF_DOF_MEASURED */
  int32_t n_meas;          /* number of measurements m (denominator of the mean, solver.py:275) */
  int32_t _pad;
} pf_mesh;

/* One material property: constant, or MLP(load_factor, x[, y]) -> softplus -> *scale
 * (fem/properties.py:97-161, examples/json/generic.py:118-142). */
typedef struct pf_net {
  int32_t enabled;     /* 0: constant property = scale */
  int32_t in_dim;      /* must be mesh.dim+1: columns [load_factor, x(, y)] (properties.py:119) */
  int32_t width;       /* hidden width h (1..32) */
  int32_t n_hidden;    /* hidden layers (1..3) */
  int32_t positive;    /* softplus on the output (enforce_positive) */
  float scale;         /* output scale = base property value */
  int32_t theta_off;   /* offset of this net's first parameter in the flat theta vector */
  int32_t pad_off;     /* offset of this net in the padded-parameter workspace */
} pf_net;

/* Device-resident iteration state; written only by kernels.  One per solve_gd call. */
typedef struct pf_state {
  int32_t iter;        /* completed iterations (= Adam step count) */
  int32_t done;        /* 1 once the stop test fired (solver.py:341-355); later launches no-op */
  int32_t converged;   /* same as done (kept separate for max_iterations exits) */
  int32_t theta_half;  /* which half holds the current theta / Adam moments: 0 = p->theta, m_t, v_t; 1 = p->theta_alt
                          (only the iteration graph ever leaves it at 1, and only between its own kernels) */
  /* Adam scalars for the NEXT step, computed in double like torch does on the host */
  float step_size_u, step_size_t, bc2_sqrt, _pad2;
  /* last iteration's monitors (solver.py:304-320) */
  float loss_total, loss_physics, loss_data, u_norm, residual_norm, theta_norm;
  int32_t u_half;      /* 1: the current displacements are in p->u_alt (only between the kernels of an iteration graph) */
  float _pad3;
} pf_state;

#define PF_HIST_COLS 6 /* loss_total, loss_physics, loss_data, u_norm, residual_norm, theta_norm */

/* Everything one GD iteration touches.  Filled by the host once per solve_gd call. */
typedef struct pf_problem {
  pf_mesh mesh;
  pf_net net[2];           /* 0: young, 1: area (density is never evaluated, nn_assembly.py:207) */
  /* unknowns + Adam moments (solver.py:205-216, 234-236); all dev */
  float* u;                /* [n_dofs] */
  float* m_u;
  float* v_u;
  float* theta;            /* [n_theta] flat, torch parameters() order young->area->density */
  float* m_t;
  float* v_t;
  int32_t n_theta;         /* all parameters incl. density net */
  int32_t n_theta_active;  /* leading parameters that receive gradients (young+area nets) */
  const int32_t* tensor_off; /* dev [n_tensors+1] parameter-tensor boundaries for theta_norm */
  int32_t n_tensors;
  int32_t wg_mode;         /* PF_WG_* : which MLP engine runs the net kernels */
  /* scalars of SolverConfig (solver.py:35-62) */
  float lam;               /* load factor of this increment */
  float alpha_physics, alpha_data;
  float lr_u, lr_t;
  double tol;
  double beta1, beta2, eps;
  int32_t use_data;        /* has_measurements && alpha_data>0 (solver.py:273) */
  int32_t max_iter;
  /* workspaces, all dev; sizes: theta_pad pf_net_pad_count() per net, partials pf_partials_count() */
  float* theta_pad;        /* padded parameter image the net kernels read */
  float* prop_e;           /* [n_elems] young per element */
  float* prop_a;           /* [n_elems] area per element */
  float* g_f;              /* [n_dofs] dL/df_int (alpha_p*r on free dofs, 0 on fixed) */
  float* g_ea;             /* [n_elems] dL/d(E*A) per element */
  float* grad_u;           /* [n_dofs] or NULL (only needed by the autograd binding) */
  float* grad_theta;       /* [n_theta] reduced parameter gradient */
  float* partials;         /* block partial sums (scalars + padded weight gradients) */
  float* hist;             /* [max_iter][PF_HIST_COLS] per-iteration monitors */
  pf_state* state;
  int32_t n_part_blocks;   /* grid size used for partial sums (host picks, <= PF_MAX_BLOCKS) */
  int32_t pad_total;       /* floats of the padded-parameter image of all active nets */
  const int32_t* pad_index;/* dev [n_theta_active] torch-layout index -> padded-image index */
  float n_meas_f;          /* (float)mesh.n_meas; multi-GPU: the GLOBAL count */
  int32_t fe_mode;         /* PF_FE_* : how fe = ke @ u_e is evaluated */
  /* multi-GPU shard interface (NULL / 0 on one GPU): the dofs of this rank's INTERFACE nodes = nodes shared with
   * other ranks plus the other nodes of the elements around them (ghost nodes here, or boundary-near nodes of which
   * another rank holds a ghost copy).  Their grad_u is completed by the iteration's all-reduce. */
  const int32_t* shared_dofs; /* dev [n_shared] local dof index */
  const int32_t* shared_slot; /* dev [n_shared] position in the global interface vector */
  int32_t n_shared;
  int32_t n_iface;         /* length of the global interface vector */
  /* multi-GPU: the local mesh = this rank's OWN elements plus the ghost elements (other ranks' elements incident to a
   * shared node), sorted by global id, so the own elements are the local range [own_lo, own_hi); 0, 0 on one GPU.
   * Gradients (MLP backward, grad_u) are taken over the own elements only. */
  int32_t own_lo, own_hi;
  int32_t part_half;        /* 0|1: which half of the residual's block partial sums this launch writes / the stand-alone
                               pf_finalize reads (0 unless the caller pipelines finalize into the next residual) */
  /* != 0: prop_e and prop_a hold 2*n_elems floats; the iteration graph then alternates between the two
   * halves, so the forwards of iteration t+1 need not wait for the last reader of iteration t's properties */
  int32_t prop_double;
  /* MFMA32 engine: split-f16 operand image of the enabled nets (pf_net_op_count() floats each, at op_off[k]),
   * rebuilt from theta by pf_pack_theta and by every theta update; NULL unless wg_mode == PF_WG_MFMA32 */
  float* net_op;
  int32_t op_off[2];
  /* 2^coord_exp * max |centroid coordinate| <= 2^14: scale of the coordinates inside the f16 gradient products */
  int32_t coord_exp;
  /* MFMA32 engine, precision of the hidden-layer and gradient matrix products: PF_MLP_F32 (2-way split f16
   * operands, float32-grade; default) or PF_MLP_BF16 (plain bf16 operands, f32 accumulate) */
  int32_t mlp_dtype;
  /* per element, the three distinct entries of ke = s*pattern, s = (E*A)/l0: [n_elems][3] = s*c2, s*cs, s*s2 (2-D),
   * [n_elems] = s (1-D); twice that with prop_double.  Written by the MFMA32 forward pass (the launch that evaluates the
   * LAST enabled net, or the fused launch of both) with the float operations of nn_assembly.py:74, 84-94, and read by the
   * node kernels instead of E, A and the geometry record (12 B per incidence instead of 20); NULL: the node kernels form
   * the same numbers from prop_e / prop_a / the constant properties and mesh.egeo */
  float* elem_k;
  /* second half of the parameter state: [3][n_theta] floats = theta | m_t | v_t, or NULL.  With it the iteration graph
   * folds the parameter update of iteration t (sum of the second-level partial rows, optimizer_theta.step(),
   * solver.py:293-294, operand images) into the prologue of the forward launch of iteration t+1: EVERY block of that
   * launch computes the update for itself from the half the previous update wrote and builds the operand images in its
   * own LDS; block 0 stores the new state into the OTHER half (nobody reads that one during the launch) and flips
   * state->theta_half.  The single-block update launch and one kernel boundary leave the critical path; a replay ends
   * with a stand-alone update that brings the state back to half 0. */
  float* theta_alt;
  /* second displacement vector [n_dofs], or NULL.  With it the iteration graph lets the displacement update of
   * iteration t (pf_node_gradu + Adam) WRITE the other vector than the one the residual and the element adjoint of t
   * read, so the update runs beside the whole backward launch instead of behind the adjoint's last read of u; the
   * vectors swap roles every iteration, an even number of iterations per replay ends in p->u, and a replay cut short by
   * the stop test copies the result home (state->u_half). */
  float* u_alt;
  /* per CSR entry of mesh.adj: the node at the OTHER end of that element ([adj_ptr[n_nodes]] int32).  With it the node
   * kernels fetch a neighbour's values one dependent load earlier (adj -> value instead of adj -> conn -> value) and
   * two incidences at a time; NULL: they go through mesh.conn. */
  const int32_t* adj_other;
} pf_problem;

#define PF_MAX_BLOCKS 1024
/* node-parallel kernels run one node per thread on up to PF_MAX_NODE_BLOCKS blocks of 256 threads;
 * each block owns one slot of the three scalar partial-sum arrays (PF_NODE_SLOTS floats each, one
 * extra slot for the multi-GPU interface dofs) at the head of `partials` */
#define PF_MAX_NODE_BLOCKS 4096
#define PF_NODE_SLOTS (PF_MAX_NODE_BLOCKS + 8)

/* ---- introspection ------------------------------------------------------------------ */
int pf_abi_version(void);
const char* pf_last_error(void);
/* number of parameters of an MLP in torch parameters() order (generic.py:121-134) */
int pf_net_param_count(int in_dim, int width, int n_hidden);
/* padded width the kernels use for `width`, or PF_ERR_UNSUPPORTED */
int pf_padded_width(int width);
/* floats of padded-parameter workspace one net needs, or <0 */
int pf_net_pad_count(int in_dim, int width, int n_hidden);
/* padded-image index (inside the net's image) of the net's `local`-th torch parameter */
int pf_net_pad_index(int in_dim, int width, int n_hidden, int local);
/* sizeof of the ABI structs: 0 pf_mesh, 1 pf_net, 2 pf_state, 3 pf_problem, 4 pf_scalar_id (binding self-check) */
int pf_sizeof(int what);
/* floats of operand-image workspace (pf_problem.net_op) one net needs with the MFMA32 engine, or <0 */
int pf_net_op_count(int in_dim, int width, int n_hidden);
/* floats needed in pf_problem.partials for the given block count */
long long pf_partials_count(const pf_problem* p);
/* which launches of one GD iteration are fused for this problem (MFMA32 engine; bit mask): 1 the forward pass of both nets
 * is ONE launch, 2 both backward passes (with the element adjoint) are ONE two-phase launch, 4 the iteration graph folds
 * the parameter update into the next forward launch, 8 the iteration graph ping-pongs two displacement vectors (the
 * displacement update on a side branch), 16 the iteration graph folds the displacement update (dL/du + Adam(u) + clamp)
 * into the next forward launch as well and is ONE chain of four launches per iteration (8 and 16 exclude each other).  In the
 * per-slot times of pf_gd_iterations_timed a fused launch is booked on the FIRST of its two slots. */
#define PF_FUSED_FORWARD 1
#define PF_FUSED_BACKWARD 2
#define PF_FUSED_THETA_UPDATE 4
#define PF_FUSED_U_PINGPONG 8
#define PF_FUSED_U_UPDATE 16
int pf_fusion_info(const pf_problem* p);

/* ---- building blocks (each replaces the cited reference lines) ------------------------ */
/* torch-layout theta -> padded image (no reference analogue; internal layout change) */
int pf_pack_theta(const pf_problem* p, void* stream);
/* NNProperty.value for every element: properties.py:116-161 + generic.py:141 (batch of n_elems
 * instead of n_elems batch-1 calls).  which: 0 young, 1 area.  Output p->prop_e / prop_a. */
int pf_net_forward(const pf_problem* p, int which, void* stream);
/* the forward pass of EVERY enabled net (one launch where pf_fusion_info says so) incl. the stiffness records, and the
 * backward pass of every enabled net with the element adjoint (one two-phase launch where fused; else adjoint + one
 * launch per net): what one GD iteration enqueues for these two steps.  Idempotent: may be repeated for timing. */
int pf_net_forward_all(const pf_problem* p, void* stream);
int pf_net_backward_all(const pf_problem* p, void* stream);
/* f_int = K(theta) u by node-wise gather of fe = (s*pattern) @ u_elem:
 * nn_assembly.py:64-100 + 226-227 (f_int only; K is never formed).  f_int_out dev [n_dofs]. */
int pf_internal_force(const pf_problem* p, const float* u, float* f_int_out, void* stream);
/* residual + losses + dL/df_int: solver.py:267-283.  Fuses pf_internal_force.  Writes p->g_f and
 * block partials of sum r^2, sum d^2; f_int_out may be NULL. */
int pf_node_residual(const pf_problem* p, float* f_int_out, void* stream);
/* per-element dL/d(EA): the part of loss.backward() (solver.py:289) through ke = s*pattern,
 * fe = ke@u (nn_assembly.py:96-100).  Writes p->g_ea. */
int pf_elem_adjoint(const pf_problem* p, void* stream);
/* backward of NNProperty.value for every element + reduction over elements of the parameter
 * gradient (autograd of properties.py:150-156 / generic.py:141).  Writes block partials. */
int pf_net_backward(const pf_problem* p, int which, void* stream);
/* dL/du by node-wise gather (K^T g_f) + data term (solver.py:273-279 backward).
 * fuse_adam!=0: also optimizer_u.step() + u[fixed]=0 (solver.py:292, 297-298) in the same pass
 * and block partials of ||u_free||^2 (solver.py:304). */
int pf_node_gradu(const pf_problem* p, int fuse_adam, void* stream);
/* sum the parameter-gradient partials in fixed order -> p->grad_theta; fuse_adam!=0: also
 * optimizer_theta.step() (solver.py:293-294) and refresh of the padded image. */
int pf_theta_reduce(const pf_problem* p, int fuse_adam, void* stream);
/* monitors, history row, stop test, Adam scalars of the next step: solver.py:304-355. */
int pf_finalize(const pf_problem* p, void* stream);
/* reset state for a new solve_gd call (fresh Adam: solver.py:234-238): zero moments, iter=0 */
int pf_reset(const pf_problem* p, void* stream);

/* ---- the fused loop ------------------------------------------------------------------- */
/* Enqueue n_iter complete GD iterations (solver.py:254-355).  Once the device-side stop test
 * fires, remaining launches are no-ops, so the final state equals the reference's `break`. */
int pf_gd_iterations(const pf_problem* p, int n_iter, void* stream);
/* hipGraph form of pf_gd_iterations: capture `iters_per_graph` iterations once, replay many times
 * (removes the per-launch host cost and most of the inter-kernel gaps).  The graph bakes in the
 * pf_problem record by value: create it after the record is final (one per solve_gd call) and destroy
 * it before changing any field.  *graph_out is an opaque handle owned by the caller.  For >= 2e5 elements the
 * captured iterations are a dependency DAG, not a chain: grad_u + Adam(u) runs beside the second net's backward, and
 * the bookkeeping of iteration t (pf_finalize's work) runs as one extra block of the residual launch of t+1, reading
 * the other half of the residual's partial sums (part_half); the last iteration of a replay gets a stand-alone
 * pf_finalize.  Results are bit-identical to pf_gd_iterations. */
int pf_graph_create(const pf_problem* p, int iters_per_graph, void* stream, void** graph_out);
/* Replays that hand their last iteration's tail to the next replay (one-chain form of the graph only, even
 * iters_per_graph; PF_ERR_UNSUPPORTED otherwise).  A plain replay ends with three stand-alone launches — the parameter
 * update, the displacement update and the bookkeeping of its last iteration (~45 us at 10^6 elements).  PF_GRAPH_NO_TAIL
 * leaves them PENDING behind the last gradient-row reduction; PF_GRAPH_CONT_HEAD makes iteration 0 carry the pending work of
 * the replay before it (as every other iteration of a replay carries its predecessor's).  Sequence: [NO_TAIL] then any
 * number of [CONT_HEAD | NO_TAIL], then pf_graph_tail (eager launches) before anything reads the state; a stop raised on
 * the device is honoured as in a plain replay (solver.py:341-355 semantics: the final state is that of the stopping
 * iteration). */
#define PF_GRAPH_CONT_HEAD 1
#define PF_GRAPH_NO_TAIL 2
int pf_graph_create_ex(const pf_problem* p, int iters_per_graph, int flags, void* stream, void** graph_out);
int pf_graph_tail(const pf_problem* p, int iters_per_graph, void* stream);
int pf_graph_launch(void* graph, void* stream);
int pf_graph_destroy(void* graph);
/* Profiling twin of pf_gd_iterations: same launches with a HIP event before every kernel slot, then
 * a stream synchronise; ms_per_kernel (HOST, PF_KERNEL_SLOTS floats) receives the average duration
 * of each slot: 0 net_forward(young) 1 net_forward(area) 2 node_residual 3 elem_adjoint
 * 4 net_backward(young) 5 net_backward(area) 6 node_gradu+Adam 7 theta_reduce+Adam 8 finalize. */
#define PF_KERNEL_SLOTS 9
int pf_gd_iterations_timed(const pf_problem* p, int n_iter, void* stream, float* ms_per_kernel);
/* loss and gradients only (no optimiser): what torch.autograd.Function.forward/backward need.
 * Requires p->grad_u != NULL.  Leaves loss terms in p->state, gradients in grad_u/grad_theta. */
int pf_loss_and_grads(const pf_problem* p, void* stream);

/* ---- multi-GPU (elements sharded across ranks) ----------------------------------------------------
 * No reference analogue: the reference is single-process.  A rank holds its own elements plus one ring of GHOST
 * elements (every element of another rank that touches a node shared with this rank), so the internal force of every
 * node of an own element is complete without communication.  One sharded iteration is
 *   pf_shard_forward          the nets on all local elements (own + ghost)
 *   pf_shard_backward         residual + losses, nets backward over the OWN elements, theta reduction into
 *                             p->grad_theta (which the host points inside buf), this rank's share of grad_u on the
 *                             interface dofs, local sums
 *                             -> buf = [iface grad_u (n_iface) | grad_theta (n_theta_active) | r2, d2, u2 of the previous iteration]
 *   pf_shard_update_interior  grad_u + Adam(u) + clamp of every dof that is NOT an interface dof
 *                                                                                      => ONE all-reduce(buf), issued by the host
 *   pf_shard_update_shared    Adam(theta) from the reduced gradient, Adam(u) of the interface dofs (every rank that
 *                             holds a copy applies the same step), u2_local[0] = this rank's sum u_free^2 over its
 *                             dofs, then the bookkeeping of the iteration (monitors, history row, stop test) from the
 *                             reduced sums; the u-norm, a monitor only, lags one iteration behind
 * and after the last iteration of a chunk: all-reduce(u2_local) => pf_shard_flush completes the last history row. */
int pf_shard_forward(const pf_problem* p, void* stream);
int pf_shard_backward(const pf_problem* p, float* buf, const float* u2_local, void* stream);
int pf_shard_update_interior(const pf_problem* p, void* stream);
int pf_shard_update_shared(const pf_problem* p, const float* buf, float* u2_local, void* stream);
int pf_shard_flush(const pf_problem* p, const float* u2_reduced, void* stream);

/* The same iteration driven from C with an own RCCL communicator (pf_comm.hip): the product path on real multi-GPU
 * runs.  librccl is dlopen'ed from `librccl_path` (the library PyTorch loaded; NULL/"" = "librccl.so"); there is no
 * link-time dependency.
 *   rank 0: pf_comm_unique_id -> 128 bytes, broadcast by the host (torch.distributed) -> every rank:
 *   pf_comm_create (collective, like ncclCommInitRank).
 * pf_shard_iterations enqueues n_iter iterations on `stream` (the four calls above around ONE ncclAllReduce each) plus
 * the closing flush, and returns without waiting for the device.  buf = dev [n_iface + n_theta_active + 3],
 * u2_local = dev [1], and p->grad_theta must point at buf + n_iface. */
#define PF_COMM_ID_BYTES 128
int pf_comm_unique_id(const char* librccl_path, void* id_out);
int pf_comm_create(const char* librccl_path, const void* id, int rank, int world, void** comm_out);
int pf_comm_destroy(void* comm);
/* failure path: ncclCommAbort instead of ncclCommDestroy — does not wait for outstanding collectives, so a rank that
 * raised can leave while its peers sit in (or it has itself enqueued) a collective that will never be matched.  Do not
 * synchronise the device first. */
int pf_comm_abort(void* comm);
/* rank and rank count as the communicator reports them (ncclCommUserRank, ncclCommCount) */
int pf_comm_info(void* comm, int* rank_out, int* nranks_out);
/* sum over ranks of buf[0..n), in place, on `stream` (the collective pf_shard_iterations issues, on its own) */
int pf_comm_all_reduce(void* comm, float* buf, int n, void* stream);
int pf_shard_iterations(const pf_problem* p, void* comm, int n_iter, float* buf, float* u2_local, void* stream);
/* iters_per_graph sharded iterations as ONE hipGraph with the collective captured inside (handle for pf_graph_destroy):
 * the single-engine iteration's dependency shape (grad_u + Adam(u) of the interior dofs beside the second backward and
 * the collective) without a host launch per kernel.  It bakes in *p, buf, u2_local and the communicator; every rank
 * must create and replay it alike.  pf_shard_iterations_graph = pf_shard_iterations replaying that graph for whole
 * multiples of iters_per_graph and launching the remainder eagerly; results are bit-identical to pf_shard_iterations. */
int pf_shard_graph_create(const pf_problem* p, void* comm, int iters_per_graph, float* buf, float* u2_local, void* stream,
                          void** graph_out);
int pf_shard_iterations_graph(const pf_problem* p, void* comm, void* graph, int iters_per_graph, int n_iter, float* buf,
                              float* u2_local, void* stream);

/* ---- classical Newton-Raphson support (pf_pcg.hip) ---------------------------
 * The reference's solve_nr (FEM/python/fem/solver.py:408-512) solves K_ff du_f = rhs_f with
 * np.linalg.solve on the dense float64 tangent of fem/assembly.py:16-75.  Here K is never formed:
 * pf_kv_f64 applies it matrix-free in float64 and pf_pcg_* run a conjugate-gradient solve preconditioned
 * with diag(K_ff) (Jacobi).  Vectors are double [n_dofs], fixed dofs carried as zeros; E and A come from
 * p->prop_e / p->prop_a when the net is enabled, else from net.scale (scalar materials). */
/* out = K v (all rows; zero_fixed != 0: rows of fixed dofs set to 0) */
int pf_kv_f64(const pf_problem* p, const double* v, double* out, int zero_fixed, void* stream);
/* doubles of workspace pf_pcg_* need */
long long pf_pcg_workspace_count(const pf_problem* p);
/* start a solve of K_ff x = b (entries of b on fixed dofs are ignored), x = 0; stop when |r| <= rtol*|b| */
int pf_pcg_begin(const pf_problem* p, const double* b, double* x, double* ws, double rtol, void* stream);
/* n_iter CG iterations (no-ops after the stop test fired).  state_out: host double[4] or NULL =
 * [iterations done, stopped (0/1), |r|^2, |b|^2], read back after a stream synchronisation. */
int pf_pcg_iterations(const pf_problem* p, double* x, double* ws, int n_iter, double* state_out, void* stream);
/* the same n_iter iterations captured as one hipGraph (handle for pf_graph_launch / pf_graph_destroy), and the
 * state read-back on its own */
int pf_pcg_graph_create(const pf_problem* p, double* x, double* ws, int n_iter, void* stream, void** graph_out);
int pf_pcg_state(const pf_problem* p, double* ws, double* state_out, void* stream);

/* ---- scalar (E, A) identification: the device loop of pinn_inverse_problem_gd -------------------------------------
 * FEM/python/api_pinn_gradient_descent.py:102-121 calls pinn_inverse_problem_gd(nodes, elements, f_ext, fixed_dofs,
 * young_init, area_init, u_measured, measured_dofs, n_iterations, learning_rate, alpha, beta) — a callee the reference
 * never defines (ImportError at :19).  The build defines it after the nearest existing code, fem/nn_solver_gd.py:105-125:
 *   loss = alpha * mean(r_free^2) + beta * mean((u_meas - u[md])^2),   r = c K_1 u - f_ext / (E0 A0),   c = exp(p_E + p_A)
 * with E = E0 exp(p_E), A = A0 exp(p_A), Adam on u (lr_u = p->lr_u) and on (p_E, p_A) (lr_p), u[fixed] = 0 after the step,
 * optional box bounds on p.  `p` describes the unit-stiffness problem (both nets disabled, scale 1, measurements in the
 * mesh, alpha_physics = alpha, alpha_data = beta); K_1 u and its transpose are the node kernels of the GD path.
 * One iteration = three launches, no host synchronisation: residual + sums, displacement update, scalar update + table. */
typedef struct pf_scalar_id {
  float* p;            /* dev [2] log-multipliers (p_E, p_A), updated in place */
  float* m_p;          /* dev [2] Adam first moments */
  float* v_p;          /* dev [2] Adam second moments */
  float* table;        /* dev [n_rows][5] per iteration: loss_total, loss_physics, loss_data, p_E, p_A (after the step) */
  int32_t n_rows;      /* rows of `table` (iterations beyond are not recorded) */
  int32_t has_bounds;  /* clamp p to [lo, hi] after the step */
  float lo[2], hi[2];
  float inv_ea0;       /* 1 / (E0 * A0): the residual is scaled so that its size does not depend on the units of E */
  float lr_p;          /* learning rate of (p_E, p_A) */
  float n_free_f;      /* number of free dofs (denominator of the mean over the residual) */
  float _pad;
} pf_scalar_id;
/* enqueue n_iter iterations (state->iter counts them; pf_reset starts a run) */
int pf_scalar_gd_iterations(const pf_problem* p, const pf_scalar_id* sp, int n_iter, void* stream);

/* ---- extensions (off the default path) ------------------------------------------------ */
/* generic Adam (torch.optim.Adam single-tensor arithmetic) on a flat vector */
int pf_adam(float* param, const float* grad, float* m, float* v, int n, int step,
            double lr, double beta1, double beta2, double eps, void* stream);
/* diag(K(theta)) [n_dofs] — the Jacobi preconditioner the north-star names; the reference has
 * no such preconditioner (solver.py:113-195 is a two-phase schedule), so nothing calls this
 * by default.  Uses p->prop_e/prop_a from the last pf_net_forward. */
int pf_diag_k(const pf_problem* p, float* diag_out, void* stream);
/* k_global in coordinate format for meshes of any size (SURVEY 7.1b `assemble_coo`): the (2*dim)^2 entries
 * `k_global[g, h] += ke[a, b]` of every element (nn_assembly.py:228-229) as triplets, element after element; rows_out /
 * cols_out dev int64 [n_elems*(2*dim)^2], vals_out dev float32; duplicates are summed by the consumer (coalesce). */
int pf_coo_k(const pf_problem* p, long long* rows_out, long long* cols_out, float* vals_out, void* stream);
/* dense k_global [n_dofs][n_dofs] row-major, for assemble_system_torch's return value
 * (nn_assembly.py:228-231); small n_dofs only (<= 4096). */
int pf_dense_k(const pf_problem* p, float* k_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PINNFEM_HIP_H */
