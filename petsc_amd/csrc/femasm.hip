// femasm.hip -- C ABI (include/femasm.h) over the gfx950 assembly kernels.
//
// One plan = one slab of the structured mesh on one GPU.  The plan owns the pattern
// arrays, the RHS scratch, the interface send/recv buffers and a stream; the caller owns
// coordinates, material and outputs (host or device).  No CPU fallback exists: every
// entry point needs a HIP device and fails loudly without one.
#include "../../include/femasm.h"
#include "femasm_kernels.h"
#include "femasm_sweep.h"
#include "femasm_hex2.h"
#include "femasm_p3hex.h"
#include "femasm_enriched.h"

#include <hipcub/hipcub.hpp>
#include <rccl/rccl.h>  // types and enums only: the four entry points used are resolved at run time (no link dependency)
#include <dlfcn.h>

#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

using namespace femasm;

namespace {

thread_local std::string g_err;

int fail(int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

#define HIPCHK(call)                                                                              \
  do {                                                                                            \
    hipError_t e_ = (call);                                                                       \
    if (e_ != hipSuccess)                                                                         \
      return fail(FEMASM_ERR_HIP, "%s:%d: %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e_)); \
  } while (0)

bool is_device_ptr(const void *p) {
  if (!p) return false;
  hipPointerAttribute_t at;
  hipError_t e = hipPointerGetAttributes(&at, p);
  if (e != hipSuccess) {
    (void)hipGetLastError();  // plain host memory: clear the sticky error
    return false;
  }
  return at.type == hipMemoryTypeDevice || at.type == hipMemoryTypeManaged;
}

inline unsigned nblk(int64_t n, int per = 256) { return (unsigned)((n + per - 1) / per); }

}  // namespace

// ---------------------------------------------------------------------------------------------------------
// Static device scratch of the one entry point that allocates nothing (femasm_error_integrals_static): the
// per-workgroup partial sums of k_error_integrals, 2 doubles per workgroup of 256 elements.  Every GPU the
// library runs on holds its own copy of this array (1 MiB) until the process ends.  In host code the NAME below is
// only a stand-in for HIP_SYMBOL(): the device address is looked up once per device (static_scratch_for) and that
// pointer is what kernels and memsets get.
// ---------------------------------------------------------------------------------------------------------
constexpr int64_t STATIC_NB_MAX = 65536;  // workgroups -> at most 256 * 65536 = 16 777 216 elements (256^3) per slab
static __device__ __align__(256) double scratch[2 * STATIC_NB_MAX];

namespace {
constexpr int MAX_DEVICES = 64;
std::mutex g_scratch_mu;
double *g_scratch_ptr[MAX_DEVICES] = {};
// address of THIS device's copy of `scratch`; called with `device` current (femasm_plan_create), once per device
int static_scratch_for(int device, double **out) {
  if (device < 0 || device >= MAX_DEVICES) return fail(FEMASM_ERR_ARG, "device %d beyond the %d this build tracks", device, MAX_DEVICES);
  std::lock_guard<std::mutex> lk(g_scratch_mu);
  if (!g_scratch_ptr[device]) {
    double *ptr = nullptr;
    HIPCHK(hipGetSymbolAddress((void **)&ptr, HIP_SYMBOL(scratch)));
    if (!ptr) return fail(FEMASM_ERR_HIP, "hipGetSymbolAddress(scratch) returned a null address on device %d", device);
    g_scratch_ptr[device] = ptr;
  }
  if (out) *out = g_scratch_ptr[device];
  return FEMASM_OK;
}
// Every plan of a GPU gets a slice of that GPU's array of its own (first fit over 32 slots of 2048 workgroups' partial
// sums), so that plans driven on different streams -- two slabs on one GPU -- never write the same words.  A plan whose
// slab does not fit what is left gets none and femasm_error_integrals_static rejects it.
constexpr int SCR_SLOTS = 32;
constexpr int64_t SCR_SLOT_NB = STATIC_NB_MAX / SCR_SLOTS;
uint32_t g_scratch_used[MAX_DEVICES] = {};
int scratch_reserve(int device, int64_t nb, int *nslots) {
  const int n = (int)((nb + SCR_SLOT_NB - 1) / SCR_SLOT_NB);
  *nslots = 0;
  if (n < 1 || n > SCR_SLOTS || device < 0 || device >= MAX_DEVICES) return -1;
  std::lock_guard<std::mutex> lk(g_scratch_mu);
  const uint32_t mask = n == 32 ? 0xffffffffu : ((1u << n) - 1u);
  for (int s0 = 0; s0 + n <= SCR_SLOTS; s0++)
    if ((g_scratch_used[device] & (mask << s0)) == 0) {
      g_scratch_used[device] |= mask << s0;
      *nslots = n;
      return s0;
    }
  return -1;
}
void scratch_release(int device, int slot, int nslots) {
  if (slot < 0 || nslots <= 0 || device < 0 || device >= MAX_DEVICES) return;
  std::lock_guard<std::mutex> lk(g_scratch_mu);
  const uint32_t mask = nslots == 32 ? 0xffffffffu : ((1u << nslots) - 1u);
  g_scratch_used[device] &= ~(mask << slot);
}
}  // namespace

namespace {
// RCCL entry points of femasm_interface_exchange, looked up once: first among the libraries already in the process (an MPI
// driver linked against librccl, or torch's bundled copy), then by loading librccl.so
struct RcclApi {
  decltype(&ncclSend) send = nullptr;
  decltype(&ncclRecv) recv = nullptr;
  decltype(&ncclGroupStart) group_start = nullptr;
  decltype(&ncclGroupEnd) group_end = nullptr;
  decltype(&ncclGetErrorString) err = nullptr;
  bool ok = false;
};
const RcclApi &rccl_api() {
  static const RcclApi api = [] {
    RcclApi a;
    void *h = RTLD_DEFAULT;  // (a null handle: "every library already in the process")
    bool have = dlsym(h, "ncclSend") != nullptr;
    if (!have) {
      h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
      if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
      have = h != nullptr;
    }
    if (have) {
      a.send = (decltype(a.send))dlsym(h, "ncclSend");
      a.recv = (decltype(a.recv))dlsym(h, "ncclRecv");
      a.group_start = (decltype(a.group_start))dlsym(h, "ncclGroupStart");
      a.group_end = (decltype(a.group_end))dlsym(h, "ncclGroupEnd");
      a.err = (decltype(a.err))dlsym(h, "ncclGetErrorString");
      a.ok = a.send && a.recv && a.group_start && a.group_end;
    }
    return a;
  }();
  return api;
}
}  // namespace

struct femasm_plan {
  int device = 0;
  int kind = 0;
  Slab s{};
  int64_t cke = 0;
  const double *coords_user = nullptr;  // as given (host or device)
  double *coords_dev = nullptr;         // device copy if coords_user is host memory
  const double *coords = nullptr;       // device pointer used by kernels
  int64_t ncoord = 0;

  int64_t nnodes = 0, nrows = 0, nnz = 0, nel = 0;
  int64_t *rowptr = nullptr;
  int32_t *colidx = nullptr;

  // RHS scratch: per-element Fe (Ge)
  double *elemF = nullptr;
  int64_t nelp = 0;                    // nel rounded up to whole blocks of 16 elements
  int64_t elemF_count = 0;             // doubles allocated in elemF
  double *lines = nullptr;             // P3 RHS on a rectilinear mesh: the three 1-D factor lines
  // interface buffers (K rows of one node layer in CSR layout, then the b slice)
  double *send = nullptr, *recv = nullptr;   // wire buffers: PACKED K rows (k_iface_pack) then the b slice
  int64_t send_count = 0, recv_count = 0, send_nnz = 0, recv_nnz = 0;  // (counts of the packed form)
  double *sfull = nullptr;                   // full CSR rows of node layer ks-1, as the stiffness kernels write them
  int64_t sfull_nnz = 0;
  bool own_send = true, own_recv = true;

  unsigned long long *flags_host = nullptr;  // pinned: the flags' read-back is a real asynchronous copy, not a staged one
  unsigned long long *bad = nullptr;   // [0] Gauss points with det(J) <= 0, [1] elements that failed the affine test
  int fast_level = 2;                  // 3-D kinds: try the row-sweep kernel (include/femasm.h), 0 = general
  bool used_fast = false;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  // femasm_interface_exchange: the send / receive pair runs on a stream of its own, so that the assembly kernels queued
  // behind it on `stream` do not wait for the neighbours (ev_send: the send buffer is final; ev_xchg: the pair is done)
  hipStream_t comm_stream = nullptr;
  hipEvent_t ev_send = nullptr, ev_xchg = nullptr;
  bool xchg_pending = false;           // femasm_interface_apply has to wait for ev_xchg
  int ncu = 256;                       // compute units of the device
  int rhs_wg_per_cu[2] = {0, 0};       // resident workgroups per CU of k_rhs_elem_elastic_runs<false / true>
  bool rect_checked = false;           // k_check_rectilinear ran on the stream in this call (its verdict is bad[3])
  bool cond_general = false;           // launch_K: the general kernels are launched with KArgs::cond set
  bool cond_rhs = false;               // launch_rhs (E2): k_rhs_e2_fused is launched with RArgs::cond set
  bool iface_rect = false;             // femasm_assemble_interface ran the rectilinearity check: its verdict carries over
  bool iface_done = false;             // femasm_assemble_interface already filled the send buffer for the next call
  bool time_main = false;              // record ev[4], ev[5] around the dominant K kernel of the main pass
  double bytes_main = 0.0;             // its algorithmic bytes

  // staging for host-side callers
  double *stage_vals = nullptr, *stage_b = nullptr, *stage_gp = nullptr;
  int64_t stage_gp_count = 0;          // doubles allocated in stage_gp (grown, never shrunk)
  const double *gp_cur = nullptr;      // DEVICE material array of the call in progress (staged copy or the caller's pointer),
                                       // set by every entry point that launches kernels: never carried from call to call
  double *partial = nullptr;           // partial sums of femasm_error_integrals / femasm_coord_moments (grown on demand)
  int64_t partial_count = 0;
  double *stage_u = nullptr;           // femasm_error_integrals: device copy of a host solution vector
  int64_t stage_u_count = 0;
  double *static_scratch = nullptr;    // this plan's slice of this device's copy of the file-scope `scratch` array
  int static_slot = -1, static_nslots = 0;  // (femasm_error_integrals_static; null / -1: no slice was left for this plan)

  double *e2_bout = nullptr;           // E2 one-pass form: launch_K hands this b array to k_assemble_e2_rect ...
  int64_t e2_b0 = 0, e2_b1 = 0;        // ... which then wrote b of the node rows [e2_b0, e2_b1)
  bool async = false;                  // femasm_plan_set_async: assemble_system / interface_apply do not wait for the device
  bool pending = false;                // an asynchronous call has not had its flags read yet (femasm_plan_sync does it)
  bool pending_fast = false;           // ... and it used a fast kernel (premise flag bad[1] decides the plan's level)

  // femasm_plan_set_option: kernel variants kept for A/B measurements and for the tests that keep them covered.
  // All zero = the automatic choice; nothing here is read from the environment.
  struct Options {
    int64_t zc = 0;              // node layers per workgroup of the row-sweep / P3 rect / S3 hexahedron kernels
    int64_t s3_units = 0;        // x-consecutive units per wave of k_assemble_s3_unit
    int64_t e2_yc = 0;           // node rows per wave of k_assemble_e2_rect
    int64_t rhs_grid = 0;        // cap on the resident grid of the S3 RHS pass
    int64_t enr_scratch_mb = 0;  // enriched kind, two-kernel form: ring size
    int s3_rowsweep = 0;         // elastic boxes: k_assemble_rowsweep<ELASTIC> instead of k_assemble_s3_unit
    int p3_sweep = 0;            // P3 general: 1 = k_assemble_sweep_p3 (round 1), 2 = k_assemble_ke_p3 (round 2) instead of k_assemble_p3_hex
    int e2_lane_loads = 0;       // E2 level 2: every lane loads its two elements' records itself (round 2) instead of the stream
    int e2_brick = 0;            // E2 general: the brick kernel for every row instead of k_assemble_ke_e2
    int e2_two_pass = 0;         // E2 boxes: K and b in two passes
    int p3_hex2w = 0;            // P3 general: the two-wave form of k_assemble_p3_hex instead of the four-wave form
    int rhs_simple = 0;          // P3 general: the load per Gauss point as the reference states it (library sinpi)
    int enr_rhs_simple = 0;      // enriched kind: the per-element RHS
    int debug = 0;               // prints launch facts to stderr
  } opt;

  // config 5 (KIND_ENRICHED)
  int n_enr = 0;                       // enrichment families per node: dof = 3 (1 + n_enr)
  int enr_mfma = -1;                   // contraction of k_enr_elem: 0 VALU, 1 FP64 MFMA (n_enr == 1 only), -1 default
  double *ke_ring = nullptr;           // element-matrix ring: ke_nring element layers
  int64_t ke_ring_count = 0;
  int ke_nring = 0;
};

namespace {

// Read the flags an asynchronous femasm_assemble_system left behind (waits for the stream): geometry errors and the
// verdict of the fast kernels' premise -- K itself is already final in stream order (the general kernels were queued
// behind the fast ones and ran on the device if the premise failed), only the plan's level is updated here.
int finish_pending(femasm_plan *p) {
  if (!p->pending) return FEMASM_OK;
  p->pending = false;
  unsigned long long *flags = p->flags_host;
  flags[0] = flags[1] = flags[2] = 0;
  HIPCHK(hipMemcpyAsync(flags, p->bad, 3 * sizeof(unsigned long long), hipMemcpyDeviceToHost, p->stream));
  HIPCHK(hipStreamSynchronize(p->stream));
  if (p->pending_fast && flags[1]) p->fast_level = (p->kind == KIND_ELASTIC && p->fast_level == 2) ? 1 : 0;  // as the synchronous loop
  p->pending_fast = false;
  if (flags[2]) return fail(FEMASM_ERR_HIP, "internal: %llu wave hand-shakes of the row-sweep kernel timed out", flags[2]);
  if (flags[0]) return fail(FEMASM_ERR_GEOMETRY, "%llu Gauss points with non-positive or NaN det(J) (reported by the "
                            "asynchronous assembly before this call)", flags[0]);
  return FEMASM_OK;
}

int traits_of(int kind, int &nsd, int &dof, int &nen, int &ngp, int &nfe, int n_enr = 0) {
  switch (kind) {
    case KIND_ENRICHED: nsd = 3; dof = 3 * (1 + n_enr); nen = 8; ngp = 8; nfe = 8 * dof; return 0;
    case KIND_POISSON: nsd = 3; dof = 1; nen = 8; ngp = 8; nfe = 8; return 0;
    case KIND_PLANE:   nsd = 2; dof = 2; nen = 4; ngp = 4; nfe = 8; return 0;
    case KIND_ELASTIC: nsd = 3; dof = 3; nen = 8; ngp = 8; nfe = 48; return 0;
  }
  return 1;
}

// number of stored entries of the rows of node layers [l0, l1) of the slowest direction
int64_t layer_nnz(const Slab &s, int64_t l0, int64_t l1) {
  const int64_t a = (s.nsd == 3) ? pairs_before(s, 0, 0, l0) : pairs_before(s, 0, l0, 0);
  const int64_t b = (s.nsd == 3) ? pairs_before(s, 0, 0, l1) : pairs_before(s, 0, l1, 0);
  return (int64_t)s.dof * s.dof * (b - a);
}

struct Box { int64_t x0, x1, y0, y1, s0, s1; };

// FEMASM_MAT_ELEMENT_SOA: two device planes that cover the slab's elements from e_ks on
static const char *check_element_soa(const femasm_plan *p, const femasm_material &m) {
  const Slab &s = p->s;
  const int64_t eplane = (s.nsd == 3) ? (s.dim[0] - 1) * (s.dim[1] - 1) : (s.dim[0] - 1);
  if (!m.gp || !is_device_ptr(m.gp)) return "the per-element (E, nu) planes must be in device memory";
  if (m.e_ks > s.els) return "material starts above the first assembled element layer";
  if ((int64_t)m.stride < eplane * (s.ele - m.e_ks)) return "material stride (plane distance) is smaller than the elements held";
  return nullptr;
}

template <int KIND>
KArgs make_kargs(femasm_plan *p, const femasm_material *mat, int bc, double *out, const Box &b, int64_t base_s0,
                 int iface) {
  using T = Traits<KIND>;
  KArgs a{};
  a.coords = p->coords;
  a.gp = (mat && (mat->layout == FEMASM_MAT_AOS || mat->layout == FEMASM_MAT_ELEMENT_SOA)) ? p->gp_cur : nullptr;
  a.out = out;
  a.bad = p->bad;
  a.base_off = (T::NSD == 3) ? row_start(p->s, 0, 0, base_s0, 0) : row_start(p->s, 0, base_s0, 0, 0);
  a.cke = p->cke;
  a.s0 = b.s0; a.s1 = b.s1;
  a.x0 = b.x0; a.x1 = b.x1; a.y0 = b.y0; a.y1 = b.y1;
  a.E = mat ? mat->E : 0.0;
  a.nu = mat ? mat->nu : 0.0;
  a.mat_layout = mat ? mat->layout : FEMASM_MAT_UNIFORM;
  a.mat_stride = mat ? mat->stride : 0;
  a.bc = bc;
  a.iface = iface;
  a.cond = p->cond_general ? 1 : 0;
  return a;
}

// general (any stencil, any BC) brick kernel over a node box
template <int KIND>
int launch_brick(femasm_plan *p, const femasm_material *mat, int bc, double *out, const Box &b, int64_t base_s0,
                 int iface) {
  using T = Traits<KIND>;
  if (b.s1 <= b.s0 || b.x1 <= b.x0 || (T::NSD == 3 && b.y1 <= b.y0)) return 0;
  const KArgs a = make_kargs<KIND>(p, mat, bc, out, b, base_s0, iface);
  dim3 grid;
  const unsigned nbx = (unsigned)((b.x1 - b.x0 + T::BX - 1) / T::BX);
  if (T::NSD == 3) grid = dim3(nbx, (unsigned)((b.y1 - b.y0 + T::BY - 1) / T::BY), (unsigned)((b.s1 - b.s0 + T::BZ - 1) / T::BZ));
  else             grid = dim3(nbx, (unsigned)((b.s1 - b.s0 + T::BY - 1) / T::BY), 1);
  if (grid.y > 65535u || grid.z > 65535u) return fail(FEMASM_ERR_ARG, "mesh too large for one launch");
  hipLaunchKernelGGL(k_assemble_rows<KIND>, grid, dim3(256), 0, p->stream, p->s, a);
  HIPCHK(hipGetLastError());
  return 0;
}

// K of node layers [s0,s1) of the slab direction into `out` (whose first row is the first row of layer s0)
template <int KIND>
int launch_K(femasm_plan *p, const femasm_material *mat, int bc, double *out, int64_t s0, int64_t s1, int iface) {
  using T = Traits<KIND>;
  if (s1 <= s0) return 0;
  const int64_t M = p->s.dim[0], N = p->s.dim[1], P = p->s.dim[2];
  if constexpr (T::NSD == 2) {
    const bool timed = p->time_main && !iface;
    const bool aos2 = mat && mat->layout == FEMASM_MAT_AOS;
    const int64_t y0 = s0 > 1 ? s0 : 1, y1 = s1 < N - 1 ? s1 : N - 1;
    if (p->fast_level == 2 && !iface && M > 2 && y1 > y0) {
      // interior nodes: 2-D row sweep (rectangles, one (E,nu) per element; verified in the kernel)
      const Box b{1, M - 1, 0, 1, y0, y1};
      KArgs a = make_kargs<KIND>(p, mat, bc, out, b, s0, iface);
      a.ends = (bc == FEMASM_BC_E2_COMPRESSION) ? 1 : 0;  // the kernel also writes the Dirichlet rows of the columns i = 0, M-1
      // one-pass form: b of these node rows from the same sweep over the records (AoS records with 16-byte alignment)
      const bool no_fuse = p->opt.e2_two_pass != 0;
      if (p->e2_bout && a.ends && aos2 && !no_fuse && ((uintptr_t)p->e2_bout & 15) == 0) {
        a.bout = p->e2_bout;
        p->e2_b0 = y0;
        p->e2_b1 = y1;
      }
      a.zc = p->opt.e2_yc > 0 ? (int32_t)p->opt.e2_yc : E2W_YC;  // node rows per wave
      dim3 grid((unsigned)((b.x1 - b.x0 + RS2_TX - 1) / RS2_TX), (unsigned)((y1 - y0 + a.zc - 1) / a.zc), 1);
      if (timed) HIPCHK(hipEventRecord(p->ev[4], p->stream));
      // one-pass form: the records as one stream per element row (stride 24, 16-byte aligned array; 2048^2: 0.414 ->
      // 0.400 ms) or loaded lane by lane.  K alone reads a third of a record: lane by lane (0.359 ms against 0.414 ms)
      const bool coal = aos2 && mat->stride == 24 && ((uintptr_t)mat->gp & 15) == 0 && !p->opt.e2_lane_loads;
      if (a.bout) {
        if (coal) hipLaunchKernelGGL((k_assemble_e2_rect<true, true>), grid, dim3(64), 0, p->stream, p->s, a);
        else hipLaunchKernelGGL((k_assemble_e2_rect<true, false>), grid, dim3(64), 0, p->stream, p->s, a);
      } else {
        hipLaunchKernelGGL((k_assemble_e2_rect<false, false>), grid, dim3(64), 0, p->stream, p->s, a);
      }
      HIPCHK(hipGetLastError());
      p->used_fast = true;
      if (timed) {
        HIPCHK(hipEventRecord(p->ev[5], p->stream));
        const double nint = (double)(M - 2) * (double)(y1 - y0);
        // stores of the interior rows + E,nu at the Gauss points (the coordinates are read by the pre-pass)
        p->bytes_main = 8.0 * nint * 36.0 + (aos2 ? 16.0 * 4.0 * (double)(M - 1) * (y1 - y0 + 1) : 0.0) +
                        (a.bout ? 16.0 * 4.0 * (double)(M - 1) * (y1 - y0 + 1) + 16.0 * (double)M * (y1 - y0) : 0.0);
      }
      // the four edges of the node grid (Dirichlet columns i = 0, M-1 and the rows j = 0, N-1): brick kernel
      int rc2 = 0;
      if (!a.ends) {
        rc2 = launch_brick<KIND>(p, mat, bc, out, Box{0, 1, 0, 1, y0, y1}, s0, iface);
        if (!rc2) rc2 = launch_brick<KIND>(p, mat, bc, out, Box{M - 1, M, 0, 1, y0, y1}, s0, iface);
      }
      if (!rc2 && s0 == 0) rc2 = launch_brick<KIND>(p, mat, bc, out, Box{0, M, 0, 1, 0, 1}, s0, iface);
      if (!rc2 && s1 == N && N - 1 >= s0 && N > 1) rc2 = launch_brick<KIND>(p, mat, bc, out, Box{0, M, 0, 1, N - 1, N}, s0, iface);
      return rc2;
    }
    if (KIND == KIND_PLANE && !iface && M > 2 && y1 > y0 && !p->opt.e2_brick) {
      // general quadrilaterals: interior nodes through the element-matrix kernel, the four edges of the node grid through
      // the brick kernel (FEMASM_E2_BRICK=1: the brick kernel for everything, as in round 1, for A/B)
      const Box b{1, M - 1, 0, 1, y0, y1};
      const KArgs a = make_kargs<KIND>(p, mat, bc, out, b, s0, iface);
      dim3 grid((unsigned)((b.x1 - b.x0 + KE2_TX - 1) / KE2_TX), (unsigned)((y1 - y0 + KE2_TY - 1) / KE2_TY), 1);
      if (timed) HIPCHK(hipEventRecord(p->ev[4], p->stream));
      hipLaunchKernelGGL(k_assemble_ke_e2, grid, dim3(128), 0, p->stream, p->s, a);
      HIPCHK(hipGetLastError());
      if (timed) {
        HIPCHK(hipEventRecord(p->ev[5], p->stream));
        const double nint = (double)(M - 2) * (double)(y1 - y0);
        p->bytes_main = 8.0 * nint * 36.0 + 8.0 * T::NSD * p->nnodes + (aos2 ? 16.0 * T::NGP * p->nel : 0.0);
      }
      int rc2 = launch_brick<KIND>(p, mat, bc, out, Box{0, 1, 0, 1, y0, y1}, s0, iface);
      if (!rc2) rc2 = launch_brick<KIND>(p, mat, bc, out, Box{M - 1, M, 0, 1, y0, y1}, s0, iface);
      if (!rc2 && s0 == 0) rc2 = launch_brick<KIND>(p, mat, bc, out, Box{0, M, 0, 1, 0, 1}, s0, iface);
      if (!rc2 && s1 == N && N - 1 >= s0 && N > 1) rc2 = launch_brick<KIND>(p, mat, bc, out, Box{0, M, 0, 1, N - 1, N}, s0, iface);
      return rc2;
    }
    if (timed) HIPCHK(hipEventRecord(p->ev[4], p->stream));
    const int rc2 = launch_brick<KIND>(p, mat, bc, out, Box{0, M, 0, 1, s0, s1}, s0, iface);
    if (timed) {
      HIPCHK(hipEventRecord(p->ev[5], p->stream));
      p->bytes_main = 8.0 * p->nnz + 8.0 * T::NSD * p->nnodes + (aos2 ? 16.0 * T::NGP * p->nel : 0.0);
    }
    return rc2;
  } else {
    const bool shell_is_bc = (bc == FEMASM_BC_POISSON_INLOOP || bc == FEMASM_BC_S3_ALLFACES_EXACT);
    // interior nodes (full 27-neighbour stencil): z-sweep kernel
    const int64_t z0 = s0 > 1 ? s0 : 1, z1 = s1 < P - 1 ? s1 : P - 1;
    bool ends_done = false;  // the interior kernel also wrote the Dirichlet rows of the x-end nodes of its lines
    if (M > 2 && N > 2 && z1 > z0) {
      const Box b{1, M - 1, 1, N - 1, z0, z1};
      const KArgs a = make_kargs<KIND>(p, mat, bc, out, b, s0, iface);
      const bool fast = p->fast_level >= 1;  // 2: verified boxes; 1: P3 parallelepipeds, S3 any hexahedra with one (E, nu) each
      const bool timed = p->time_main && !iface;
      const bool rect = fast && p->fast_level == 2;  // level 2: geometry from three coordinate lines (launch_rect_check ran)
      if (fast && KIND == KIND_ELASTIC && !rect) {
        // level 1 of the elastic kind, pre-pass: premise test (one (E, nu) per element) + the elements' coefficients into
        // the element-vector scratch, which is free until the RHS pass (k_s3_elem_coefs in femasm_hex2.h)
        const int64_t nel_here = (M - 1) * (N - 1) * (p->s.ele - p->s.els);
        if (4 * nel_here > p->elemF_count) return fail(FEMASM_ERR_ARG, "element scratch too small for the coefficient pre-pass");
        const int64_t nb = (nel_here + 127) / 128;
        hipLaunchKernelGGL(k_s3_elem_coefs, dim3((unsigned)(nb < 65536 ? nb : 65536)), dim3(256), 0, p->stream, p->s, a, p->elemF);
        HIPCHK(hipGetLastError());
      }
      if (timed) {
        HIPCHK(hipEventRecord(p->ev[4], p->stream));
        const bool aos3 = (KIND != KIND_POISSON) && mat && mat->layout == FEMASM_MAT_AOS;
        const double nint = (double)(M - 2) * (double)(N - 2) * (double)(z1 - z0);  // interior nodes = rows / dof
        // stores of the interior rows + one read of the coordinates and of E,nu at the Gauss points
        // (at level 2 the coordinates are read by the rectilinearity pre-pass, not by this kernel)
        // (elastic level 1: the E, nu of the records are read by the pre-pass, outside this bracket; the kernel itself reads
        // the 32 bytes per element the pre-pass left, whatever the material layout)
        const bool hex2 = fast && KIND == KIND_ELASTIC && !rect;
        p->bytes_main = 8.0 * nint * 27.0 * T::DOF * T::DOF +
                        (p->fast_level == 2 ? 0.0 : 8.0 * T::NSD * (double)M * N * (z1 - z0 + 2)) +
                        (hex2 ? 32.0 * (double)(M - 1) * (N - 1) * (z1 - z0 + 1)
                              : (aos3 ? 16.0 * T::NGP * (double)(M - 1) * (N - 1) * (z1 - z0 + 1) : 0.0));
      }
      if (fast && KIND == KIND_ELASTIC && !rect) {
        // level 1 of the elastic kind: any hexahedra, the material taken out of the Gauss-point loop
        // (femasm_hex2.h: four-wave workgroups on 8 x 2 node tiles, two per CU)
        const int64_t tiles = ((b.x1 - b.x0 + H2_TX - 1) / H2_TX) * ((b.y1 - b.y0 + H2_TY - 1) / H2_TY);
        // node layers per workgroup: the chunking that wastes least -- a chunk of zc node layers sweeps zc + 1 element
        // layers, and the last round of the resident workgroups should be full
        const int64_t nz = z1 - z0, resident = (int64_t)p->ncu * 2;
        int64_t zc = nz;
        double best = -1.0;
        for (int64_t chunks = 1; chunks <= nz; chunks++) {
          const int64_t c = (nz + chunks - 1) / chunks, rc = (nz + c - 1) / c;
          if (c < 6 && chunks > 1) break;
          const int64_t W = tiles * rc, rounds = (W + resident - 1) / resident;
          // (workgroups differ in speed by 10-15 %: the fewer rounds, the more of the last one is spent waiting for the
          // slowest -- 128^3: one chunk = two rounds 1.81 ms, three chunks 1.745 ms, five 1.78 ms)
          const double eff = (double)W / (double)(rounds * resident) * (double)nz / (double)(nz + rc) * (1.0 - 0.25 / (double)rounds);
          if (eff > best + 1e-9) { best = eff; zc = c; }
        }
        if (p->opt.zc > 0) zc = p->opt.zc;
        KArgs az = a;
        az.zc = (int32_t)zc;
        az.ecoef = p->elemF;
        dim3 grid((unsigned)((b.x1 - b.x0 + H2_TX - 1) / H2_TX), (unsigned)((b.y1 - b.y0 + H2_TY - 1) / H2_TY),
                  (unsigned)((z1 - z0 + zc - 1) / zc));
        if (p->opt.debug) fprintf(stderr, "[femasm] elastic level 1: grid %u x %u x %u, %lld node layers per workgroup\n", grid.x, grid.y, grid.z, (long long)zc);
        hipLaunchKernelGGL(k_assemble_s3_hex2, grid, dim3(H2_NT), 0, p->stream, p->s, az);
        p->used_fast = true;
      } else if (fast) {
        constexpr int TX = RowSweep<KIND>::TX, TY = RowSweep<KIND>::TY;
        // node layers per workgroup.  Row-sweep kernel (2 workgroups per CU resident): of 16 / 32 / 64 the one that
        // fills the last round of resident workgroups best, weighed against the ~2-layer prologue of a workgroup
        // (128^3: 64 -> 512 workgroups = exactly one round, 0.92 ms against 0.94 ms with 32; 96^3: 32 -> 0.47 ms against
        // 0.70 ms with 64; 160^3: 32).  Autonomous-wave Poisson kernel: ONE -- every wave is a one-shot unit that
        // writes a single contiguous run, dispatched in memory order, which is the pattern the memory system takes fastest
        // (scripts/ubench/fill_roof*.hip: one-shot runs in order 5.7-6.8 TB/s, persistent strided writers 4.5-4.8);
        // 256^3: 0.62 ms against 0.73 ms with 16 layers per wave (2, 4, 8: 0.71-0.72), 128^3: 0.092 against 0.105.
        // FEMASM_ZC overrides for experiments.
        int64_t zc = 1;
        if (!(KIND == KIND_POISSON && rect)) {
          const double resident = 2.0 * p->ncu;
          const double tiles = (double)((b.x1 - b.x0 + TX - 1) / TX) * (double)((b.y1 - b.y0 + TY - 1) / TY);
          double best = -1.0;
          for (int64_t c : {16, 32, 64}) {
            const double W = tiles * (double)((z1 - z0 + c - 1) / c);
            const double score = W / (std::ceil(W / resident) * resident) * (double)c / (double)(c + 2);
            if (score > best) { best = score; zc = c; }
          }
        }
        if (p->opt.zc > 0) zc = p->opt.zc;
        if (zc > 128) zc = 128;
        KArgs az = a;
        az.zc = (int32_t)zc;
        az.rect = rect ? 1 : 0;
        if (KIND == KIND_POISSON && rect) {  // dof 1 on a rectilinear mesh: autonomous waves, one tile row each
          ends_done = shell_is_bc;
          az.ends = ends_done ? 1 : 0;
          dim3 grid((unsigned)((b.x1 - b.x0 + P3W_TX - 1) / P3W_TX), (unsigned)(b.y1 - b.y0),
                    (unsigned)((z1 - z0 + zc - 1) / zc));
          hipLaunchKernelGGL(k_assemble_p3_rect, grid, dim3(64), 0, p->stream, p->s, az);
        } else if (KIND == KIND_ELASTIC && rect && !p->opt.s3_rowsweep && b.y1 - b.y0 < 65536 && z1 - z0 < 65536) {
          // elastic on a rectilinear mesh: one-shot units (one wave = one 8-node run each) in memory order
          // (FEMASM_S3_ROWSWEEP=1: the persistent row-sweep kernel of round 1, for A/B)
          // x-consecutive units per wave: as many as leave >= 8192 waves (four rounds of the chip at two waves per SIMD);
          // at 128^3 a wave walks a whole line (measured: 1 unit 1.09 ms, 4: 1.04, 8: 1.00, 16: 0.91-0.93)
          const int64_t ux = (b.x1 - b.x0 + SU_TX - 1) / SU_TX, lines = (int64_t)(b.y1 - b.y0) * (z1 - z0);
          az.zc = (int32_t)(p->opt.s3_units > 0 ? p->opt.s3_units : std::max<int64_t>(1, std::min<int64_t>(ux, ux * lines / 8192)));
          const int64_t per = (int64_t)az.zc * SU_TX;
          dim3 grid((unsigned)((b.x1 - b.x0 + per - 1) / per), (unsigned)(b.y1 - b.y0), (unsigned)(z1 - z0));
          hipLaunchKernelGGL(k_assemble_s3_unit, grid, dim3(64), 0, p->stream, p->s, az);
        } else {
          dim3 grid((unsigned)((b.x1 - b.x0 + TX - 1) / TX), (unsigned)((b.y1 - b.y0 + TY - 1) / TY),
                    (unsigned)((z1 - z0 + zc - 1) / zc));
          hipLaunchKernelGGL(k_assemble_rowsweep<KIND>, grid, dim3(64 * (3 + RowSweep<KIND>::NRW)), 0, p->stream, p->s, az);
        }
        p->used_fast = true;
      } else {
        dim3 grid((unsigned)((b.x1 - b.x0 + SW_TX - 1) / SW_TX), (unsigned)((b.y1 - b.y0 + SW_TY - 1) / SW_TY),
                  (unsigned)((z1 - z0 + SW_ZC - 1) / SW_ZC));
        if (KIND == KIND_POISSON && !p->opt.p3_sweep && 3 * M * N < ((int64_t)1 << 31)) {
          // element matrices in tensor-product form, through LDS (femasm_p3hex.h; option p3_sweep = 2: the element-matrix
          // kernel of round 2, 1: the node-row kernel of round 1, for A/B)
          // node layers per workgroup: a chunk of zc node layers sweeps zc + 1 element layers, and the last round of the
          // resident workgroups (two per CU: one wave per SIMD) should be full (256^3: 16 -> 1.61 ms, 32 -> 1.57 ms)
          // (the default kernel: one eight-wave workgroup per CU on two x-neighbouring tiles; the two-wave form: two
          // workgroups per CU on one tile each)
          const int64_t ntx = (b.x1 - b.x0 + PH_TX - 1) / PH_TX, nty = (b.y1 - b.y0 + PH_TY - 1) / PH_TY;
          const int64_t tiles = (p->opt.p3_hex2w ? ntx : (ntx + 1) / 2) * nty;
          const int64_t nz = z1 - z0, resident = (int64_t)p->ncu * (p->opt.p3_hex2w ? 2 : 1);
          int64_t zc = nz;
          double best = -1.0;
          for (int64_t chunks = 1; chunks <= nz; chunks++) {
            const int64_t c = (nz + chunks - 1) / chunks, rc = (nz + c - 1) / c;
            if (c < 8 && chunks > 1) break;
            const int64_t W = tiles * rc, rounds = (W + resident - 1) / resident;
            const double eff = (double)W / (double)(rounds * resident) * (double)nz / (double)(nz + rc) * (1.0 - 0.25 / (double)rounds);
            if (eff > best + 1e-9) { best = eff; zc = c; }
          }
          if (p->opt.zc > 0) zc = p->opt.zc;
          KArgs az = a;
          az.zc = (int32_t)zc;
          if (p->opt.debug) fprintf(stderr, "[femasm] poisson level 0: %lld tiles, %lld node layers per workgroup\n", (long long)tiles, (long long)zc);
          dim3 gph((unsigned)((b.x1 - b.x0 + PH_TX - 1) / PH_TX), (unsigned)((b.y1 - b.y0 + PH_TY - 1) / PH_TY),
                   (unsigned)((nz + zc - 1) / zc));
          if (p->opt.p3_hex2w) hipLaunchKernelGGL(k_assemble_p3_hex, gph, dim3(PH_NT), 0, p->stream, p->s, az);
          else hipLaunchKernelGGL(k_assemble_p3_hex4, dim3((gph.x + 1) / 2, gph.y, gph.z), dim3(512), 0, p->stream, p->s, az);
        } else if (KIND == KIND_POISSON && p->opt.p3_sweep != 1) {
          dim3 gke((unsigned)((b.x1 - b.x0 + KE_TX - 1) / KE_TX), (unsigned)((b.y1 - b.y0 + KE_TY - 1) / KE_TY),
                   (unsigned)((z1 - z0 + KE_ZC - 1) / KE_ZC));
          hipLaunchKernelGGL(k_assemble_ke_p3, gke, dim3(256), 0, p->stream, p->s, a);
        } else if (KIND == KIND_POISSON) hipLaunchKernelGGL(k_assemble_sweep_p3, grid, dim3(256), 0, p->stream, p->s, a);
        else hipLaunchKernelGGL(k_assemble_sweep<KIND>, grid, dim3(256), 0, p->stream, p->s, a);
      }
      HIPCHK(hipGetLastError());
      if (timed) HIPCHK(hipEventRecord(p->ev[5], p->stream));
    }
    if (shell_is_bc) {
      // Dirichlet rows of the boundary shell, AFTER the interior kernel: this small store kernel then runs while the
      // interior kernel's dirty lines drain, and the read-bound RHS kernels that follow start on a quiet memory system
      KArgs a = make_kargs<KIND>(p, mat, bc, out, Box{0, M, 0, N, s0, s1}, s0, iface);
      a.ends = ends_done ? 1 : 0;
      const int64_t nrow = N * (s1 - s0);
      if (nrow >= ((int64_t)1 << 31)) return fail(FEMASM_ERR_ARG, "mesh too large for one launch");
      hipLaunchKernelGGL(k_bc_rows, dim3((unsigned)(nrow < 8192 ? nrow : 8192)), dim3(256), 0, p->stream, p->s, a);
      HIPCHK(hipGetLastError());
    }
    if (!shell_is_bc) {  // boundary rows that are real equations: general brick kernel on the six faces
      int rc = 0;
      if (s0 == 0) rc = launch_brick<KIND>(p, mat, bc, out, Box{0, M, 0, N, 0, 1}, s0, iface);
      if (!rc && s1 == P && P - 1 >= s0 && P > 1) rc = launch_brick<KIND>(p, mat, bc, out, Box{0, M, 0, N, P - 1, P}, s0, iface);
      if (!rc && z1 > z0) {
        rc = launch_brick<KIND>(p, mat, bc, out, Box{0, M, 0, 1, z0, z1}, s0, iface);
        if (!rc && N > 1) rc = launch_brick<KIND>(p, mat, bc, out, Box{0, M, N - 1, N, z0, z1}, s0, iface);
        if (!rc && N > 2) rc = launch_brick<KIND>(p, mat, bc, out, Box{0, 1, 1, N - 1, z0, z1}, s0, iface);
        if (!rc && N > 2 && M > 1) rc = launch_brick<KIND>(p, mat, bc, out, Box{M - 1, M, 1, N - 1, z0, z1}, s0, iface);
      }
      if (rc) return rc;
    }
    return 0;
  }
}

// Level 2, 3-D: exact test that the slab's mesh is rectilinear, once per call before the stiffness and RHS kernels
// that rely on it (flags bad[1] and bad[3]).
int launch_rect_check(femasm_plan *p) {
  p->rect_checked = false;
  if (p->fast_level != 2) return 0;
  const int64_t nlayers = p->cke - p->s.cks;
  if (p->s.nsd == 2) {
    int64_t nb2 = nlayers < (int64_t)p->ncu * 16 ? nlayers : (int64_t)p->ncu * 16;
    hipLaunchKernelGGL(k_check_rectilinear2d, dim3((unsigned)nb2), dim3(256), 0, p->stream, p->s, p->coords, nlayers,
                       p->bad + 1);
    HIPCHK(hipGetLastError());
    p->rect_checked = true;
    return 0;
  }
  int64_t nb = p->s.dim[1] * nlayers;
  if (nb > (int64_t)p->ncu * 16) nb = (int64_t)p->ncu * 16;
  hipLaunchKernelGGL(k_check_rectilinear, dim3((unsigned)nb), dim3(256), 0, p->stream, p->s, p->coords, nlayers,
                     p->bad + 1);
  HIPCHK(hipGetLastError());
  p->rect_checked = true;
  return 0;
}

// full rows of node layer ks-1 (p->sfull) -> the packed wire buffer (p->send)
int launch_iface_pack(femasm_plan *p) {
  const int64_t plane = (p->s.nsd == 3) ? p->s.dim[0] * p->s.dim[1] : p->s.dim[0];
  hipLaunchKernelGGL(k_iface_pack, dim3(nblk(plane, 4)), dim3(256), 0, p->stream, p->s, p->s.ks - 1, p->sfull, p->send);
  HIPCHK(hipGetLastError());
  return 0;
}

template <int KIND>
int launch_rhs(femasm_plan *p, const femasm_material *mat, int bc, double *out, int64_t s0, int64_t s1, int iface,
               bool do_elem, const Slab *slab = nullptr, int64_t nel = -1) {
  using T = Traits<KIND>;
  if (s1 <= s0) return 0;
  const Slab &sl = slab ? *slab : p->s;  // element layers [els, ele) to integrate / to gather from
  if (nel < 0) nel = p->nel;
  RArgs a{};
  a.coords = p->coords;
  a.gp = (mat && mat->layout == FEMASM_MAT_AOS) ? p->gp_cur : nullptr;
  a.elemF = p->elemF;
  a.out = out;
  a.cke = p->cke;
  a.s0 = s0;
  a.s1 = s1;
  a.nel = nel;
  a.mat_layout = mat ? mat->layout : FEMASM_MAT_UNIFORM;
  a.mat_stride = mat ? mat->stride : 0;
  a.bc = bc;
  a.iface = iface;
  a.bad = p->bad;
  a.rect = p->rect_checked ? 1 : 0;
  a.E = mat ? mat->E : 0.0;
  a.nu = mat ? mat->nu : 0.0;
  a.cond = p->cond_rhs ? 1 : 0;
  a.simple = p->opt.rhs_simple;
  // S3/R3 with the reference's AoS records as 16-byte runs: streaming element pass + in-wave part of the node sum
  // (k_rhs_elem_elastic_runs); anything else (uniform material: no loads; an odd stride or base) takes the
  // thread-per-element kernel and the element-vector form of the gather
  const int64_t nlay = sl.ele - sl.els;
  const int64_t ntasks = ((nlay + RUNS_ZC - 1) / RUNS_ZC) * (p->s.dim[1] - 1) * ((p->s.dim[0] - 1 + 15) / 16);
  const bool runs = KIND == KIND_ELASTIC && a.gp && (a.mat_stride & 1) == 0 && ((uintptr_t)a.gp & 15) == 0 &&
                    ntasks < ((int64_t)1 << 31) && ntasks * (RUNS_ZC + 1) * 128 <= p->elemF_count;
  a.runs = runs ? 1 : 0;
  if (KIND == KIND_POISSON && a.rect) {
    // rectilinear mesh (verdict of the pre-pass read on the device): b as an outer product of three lines; the
    // element pass and the gather below then return at once
    const int64_t nl = p->s.dim[0] + p->s.dim[1] + (p->cke - p->s.cks);
    if (!p->lines) HIPCHK(hipMalloc(&p->lines, (size_t)nl * sizeof(double)));
    hipLaunchKernelGGL(k_poisson_rhs_lines, dim3(nblk(nl)), dim3(256), 0, p->stream, sl, a, p->lines);
    if (p->s.dim[0] * p->s.dim[1] >= ((int64_t)1 << 32)) return fail(FEMASM_ERR_ARG, "node plane too large");
    const dim3 g3(nblk(p->s.dim[0] * p->s.dim[1]), (unsigned)(s1 - s0 < 64 ? s1 - s0 : 64), 1);
    hipLaunchKernelGGL(k_poisson_rhs_rect, g3, dim3(256), 0, p->stream, sl, a, p->lines);
    HIPCHK(hipGetLastError());
  }
  if constexpr (KIND == KIND_PLANE) {
    // E2: element integration and node sum in one pass (k_rhs_e2_fused), no element-vector array
    const bool live = a.gp != nullptr && a.mat_layout == FEMASM_MAT_AOS && sl.ele > sl.els && p->s.dim[0] > 1;
    const bool al = (((uintptr_t)a.gp | (uintptr_t)a.coords) & 15) == 0 && (a.mat_stride & 1) == 0;
    const int64_t nwx = (p->s.dim[0] + 62) / 63, ngy = (s1 - s0 + E2R_YC - 1) / E2R_YC;
    if (ngy > 65535) return fail(FEMASM_ERR_ARG, "mesh too large for one launch");
    const dim3 fgrid((unsigned)((nwx + 3) / 4), (unsigned)ngy, 1);
    if (!live) hipLaunchKernelGGL((k_rhs_e2_fused<false, false>), fgrid, dim3(256), 0, p->stream, sl, a);
    else if (al) hipLaunchKernelGGL((k_rhs_e2_fused<true, true>), fgrid, dim3(256), 0, p->stream, sl, a);
    else hipLaunchKernelGGL((k_rhs_e2_fused<false, true>), fgrid, dim3(256), 0, p->stream, sl, a);
    HIPCHK(hipGetLastError());
    return 0;
  }
  if (do_elem && nel > 0) {
    if (runs) {
      // persistent one-wave workgroups: one resident set (occupancy queried once per plan)
      int *wg_per_cu = p->rhs_wg_per_cu;
      const int ncu = p->ncu;
      if (!wg_per_cu[0]) {
        HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&wg_per_cu[1], k_rhs_elem_elastic_runs<true>, 64, 0));
        HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&wg_per_cu[0], k_rhs_elem_elastic_runs<false>, 64, 0));
        if (wg_per_cu[0] < 1) wg_per_cu[0] = 1;
        if (wg_per_cu[1] < 1) wg_per_cu[1] = 1;
        if (p->opt.debug) fprintf(stderr, "[femasm] rhs runs: ncu %d, workgroups per CU %d (general) %d (rect)\n", ncu, wg_per_cu[0], wg_per_cu[1]);
      }
      // FEMASM_RHS_GRID caps the grid (tests: several tasks per persistent wave on small meshes)
      const int64_t grid_cap = p->opt.rhs_grid;
      auto grid = [&](int v) {
        int64_t r = (int64_t)ncu * wg_per_cu[v];
        if (grid_cap > 0 && grid_cap < r) r = grid_cap;
        return dim3((unsigned)(ntasks < r ? ntasks : r));
      };
      if (a.rect) hipLaunchKernelGGL(k_rhs_elem_elastic_runs<true>, grid(1), dim3(64), 0, p->stream, sl, a);
      hipLaunchKernelGGL(k_rhs_elem_elastic_runs<false>, grid(0), dim3(64), 0, p->stream, sl, a);
    }
    else if (KIND == KIND_ELASTIC && a.mat_layout == FEMASM_MAT_MANUFACTURED)
      hipLaunchKernelGGL(k_rhs_elem_s3_manufactured, dim3(nblk(nel, 128) < 65536u ? nblk(nel, 128) : 65536u), dim3(128), 0,
                         p->stream, sl, a);
    else
      hipLaunchKernelGGL(k_rhs_elem<KIND>, dim3(nblk(nel) < 16384u ? nblk(nel) : 16384u), dim3(256), 0, p->stream, sl, a);
    HIPCHK(hipGetLastError());
  }
  // (the gather walks its logical grid with strides in y and z: capped launch)
  auto capped = [](int64_t n, int64_t cap) { return (unsigned)(n < cap ? n : cap); };
  const dim3 ggrid = (T::NSD == 3) ? dim3(capped((p->s.dim[0] * p->s.dim[1] + 255) / 256, 1024), 1, capped(s1 - s0, 64))
                                   : dim3(capped((p->s.dim[0] * (s1 - s0) + 255) / 256, 16384), 1, 1);
  if constexpr (KIND == KIND_ELASTIC) {
    if (runs) hipLaunchKernelGGL((k_rhs_gather<KIND_ELASTIC, true>), ggrid, dim3(256), 0, p->stream, sl, a);
    else hipLaunchKernelGGL(k_rhs_gather<KIND>, ggrid, dim3(256), 0, p->stream, sl, a);
  } else {
    hipLaunchKernelGGL(k_rhs_gather<KIND>, ggrid, dim3(256), 0, p->stream, sl, a);
  }
  HIPCHK(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------------------------------------
// Config 5: assembly of the enriched kind.  Element layers are processed in chunks through a ring of element
// matrices (k_enr_elem), each chunk followed by the segmented reduction of the node layers it completes
// (k_enr_gather); then the right-hand side.
// ---------------------------------------------------------------------------------------------------------
template <int S>
void launch_enr_elem(femasm_plan *p, const EnrArgs &a, int64_t nelem, bool mfma) {
  constexpr int NW = EnrCfg<S>::NW;
  const dim3 grid((unsigned)((nelem + NW - 1) / NW));
  if constexpr (S == 2) {
    if (mfma) {
      hipLaunchKernelGGL((k_enr_elem<2, true>), grid, dim3(64 * NW), 0, p->stream, p->s, a);
      return;
    }
  }
  hipLaunchKernelGGL((k_enr_elem<S, false>), grid, dim3(64 * NW), 0, p->stream, p->s, a);
}

template <int S>
void launch_enr_node(femasm_plan *p, const EnrArgs &a, int64_t nnodes, bool onemat) {
  constexpr int per_wg = EnrNodeCfg<S>::NW * EnrNodeCfg<S>::NPW;
  const dim3 grid((unsigned)((nnodes + per_wg - 1) / per_wg));
  if (onemat) hipLaunchKernelGGL((k_enr_node<S, true>), grid, dim3(64 * EnrNodeCfg<S>::NW), 0, p->stream, p->s, a);
  else hipLaunchKernelGGL((k_enr_node<S, false>), grid, dim3(64 * EnrNodeCfg<S>::NW), 0, p->stream, p->s, a);
}
template <int S>
void launch_enr_box(femasm_plan *p, const EnrArgs &a0, int64_t nnodes) {
  // consecutive nodes per wave: as many as leave eight rounds of workgroups (one per CU), 4 .. 32 (a wave's first node
  // has its element table built in the open: 128^3, n_enr = 1: 4 -> 8.55 ms, 8 -> 8.23, 16 -> 8.12, 32 -> 8.04)
  constexpr int NW = EnrBoxCfg<S>::NW;
  int64_t npw = nnodes / ((int64_t)NW * 8 * p->ncu);
  npw = npw < 4 ? 4 : (npw > EnrBoxCfg<S>::NPW_MAX ? EnrBoxCfg<S>::NPW_MAX : npw);
  EnrArgs a = a0;
  a.npw = (int32_t)npw;
  const int64_t per_wg = NW * npw;
  const dim3 grid((unsigned)((nnodes + per_wg - 1) / per_wg));
  hipLaunchKernelGGL((k_enr_box<S>), grid, dim3(64 * NW), 0, p->stream, p->s, a);
}

int assemble_enriched(femasm_plan *p, const femasm_material *mat, int bc, double *csr_values, double *b,
                      femasm_stats *stats) {
  const auto t_begin = std::chrono::steady_clock::now();
  const Slab &s = p->s;
  const int S = 1 + p->n_enr, D = 3 * S;
  if (bc != FEMASM_BC_NONE && bc != FEMASM_BC_S3_ALLFACES_EXACT)
    return fail(FEMASM_ERR_ARG, "the enriched kind takes FEMASM_BC_NONE or FEMASM_BC_S3_ALLFACES_EXACT");
  if (!mat || mat->layout != FEMASM_MAT_AOS || !mat->gp)
    return fail(FEMASM_ERR_ARG, "the enriched kind needs the AoS records (R3 GaussPointCoefficients + enrichment blocks)");
  const int base = mat->stride - 64 * p->n_enr;
  if (base != 408 && base != 416 && !(p->n_enr == 0 && mat->stride == 400))
    return fail(FEMASM_ERR_ARG, "enriched record stride %d is not 408 + 64 n_enr (32-bit PetscInt) or 416 + 64 n_enr", mat->stride);
  if (mat->e_ks > s.els) return fail(FEMASM_ERR_ARG, "material starts above the first assembled element layer");
  if (!is_device_ptr(mat->gp) || (csr_values && !is_device_ptr(csr_values)) || (b && !is_device_ptr(b)))
    return fail(FEMASM_ERR_ARG, "the enriched kind works on device-resident records, csr_values and b");
  if (mat->enr_mode != 0 && mat->enr_mode != 1) return fail(FEMASM_ERR_ARG, "enr_mode %d: 0 = SGFEM, 1 = GFEM", mat->enr_mode);
  p->s.mat_e_ks = mat->e_ks;
  p->gp_cur = mat->gp;
  const int64_t M = s.dim[0], N = s.dim[1], P = s.dim[2];
  const int64_t eplane = (M - 1) * (N - 1), nlay = s.ele - s.els;
  const int64_t KE = 64 * (int64_t)D * D;

  EnrArgs a{};
  a.coords = p->coords;
  a.gp = mat->gp;
  a.bad = p->bad;
  a.cke = p->cke;
  a.stride = mat->stride;
  a.base = (p->n_enr == 0 && mat->stride == 400) ? 400 : base;  // (n_enr == 0 never reads the flags)
  a.sgfem = mat->enr_mode == 0 ? 1 : 0;
  a.bc = bc;
  a.elemF = p->elemF;
  HIPCHK(hipMemsetAsync(p->bad, 0, 4 * sizeof(unsigned long long), p->stream));
  if (stats) HIPCHK(hipEventRecord(p->ev[0], p->stream));

  if (csr_values && nlay > 0) {
    // ring of element layers: as many as the scratch budget holds (FEMASM_ENR_SCRATCH_MB, default 3072), at least 2
    const bool two_phase = p->enr_mfma == 0 || p->enr_mfma == 1;  // else: one node-centric pass, no ring
    int64_t budget = 3072;
    if (p->opt.enr_scratch_mb > 0) budget = p->opt.enr_scratch_mb;
    const int64_t layer_doubles = eplane * KE;
    int64_t nring = (budget << 20) / (layer_doubles * 8);
    if (nring > nlay + 1) nring = nlay + 1;
    if (nring < 2) nring = 2;
    if (two_phase && nring * layer_doubles > p->ke_ring_count) {
      if (p->ke_ring) (void)hipFree(p->ke_ring);
      p->ke_ring = nullptr;
      p->ke_ring_count = 0;
      HIPCHK(hipMalloc(&p->ke_ring, (size_t)(nring * layer_doubles) * sizeof(double)));
      p->ke_ring_count = nring * layer_doubles;
    }
    p->ke_nring = (int)nring;
    a.ke = p->ke_ring;
    a.nring = (int)nring;
    const int64_t chunk = nring - 1;
    // default: VALU.  Measured: both variants do the same 86 016 flops per element and FP64 MFMA has the
    // VALU's peak on this part, so the contraction alone is a tie (0.385 vs 0.395 ms per 147 456 elements) and the
    // kernel is bound by its stores either way (0.687 vs 0.657 ms)
    const bool mfma = (S == 2) && p->enr_mfma == 1;
    auto gather = [&](double *out, int64_t l0, int64_t l1, int iface) {
      if (l1 <= l0) return;
      EnrArgs g = a;
      g.out = out;
      g.s0 = l0;
      g.s1 = l1;
      g.iface = iface;
      g.base_off = iface ? row_start(s, 0, 0, l0, 0) : 0;  // owned rows: positions relative to the slab's first row
      const int64_t nn = M * N * (l1 - l0);
      const dim3 grid((unsigned)((nn + ENR_GW - 1) / ENR_GW));
      if (D == 3) hipLaunchKernelGGL(k_enr_gather<3>, grid, dim3(64 * ENR_GW), 0, p->stream, s, g);
      else if (D == 6) hipLaunchKernelGGL(k_enr_gather<6>, grid, dim3(64 * ENR_GW), 0, p->stream, s, g);
      else if (D == 9) hipLaunchKernelGGL(k_enr_gather<9>, grid, dim3(64 * ENR_GW), 0, p->stream, s, g);
      else hipLaunchKernelGGL(k_enr_gather<12>, grid, dim3(64 * ENR_GW), 0, p->stream, s, g);
    };
    if (M * N * (s.ke - s.ks + 1) / ENR_GW >= ((int64_t)1 << 31)) return fail(FEMASM_ERR_ARG, "slab too large for one launch");
    if (stats) HIPCHK(hipEventRecord(p->ev[4], p->stream));
    // default: one node-centric pass (k_enr_node), nothing but the CSR values leaves the chip; the two-kernel
    // form (element matrices through a scratch ring) is what femasm_plan_set_enriched_contraction(0 | 1) selects
    bool box = false, onemat = false;
    auto node_pass = [&](double *out, int64_t l0, int64_t l1, int iface) {
      if (l1 <= l0) return;
      EnrArgs g = a;
      g.out = out;
      g.s0 = l0;
      g.s1 = l1;
      g.iface = iface;
      g.base_off = iface ? row_start(s, 0, 0, l0, 0) : 0;
      const int64_t nn = M * N * (l1 - l0);
      if (box) {
        switch (S) {
          case 1: launch_enr_box<1>(p, g, nn); break;
          case 2: launch_enr_box<2>(p, g, nn); break;
          case 3: launch_enr_box<3>(p, g, nn); break;
          default: launch_enr_box<4>(p, g, nn); break;
        }
      } else {
        switch (S) {
          case 1: launch_enr_node<1>(p, g, nn, onemat); break;
          case 2: launch_enr_node<2>(p, g, nn, onemat); break;
          case 3: launch_enr_node<3>(p, g, nn, onemat); break;
          default: launch_enr_node<4>(p, g, nn, onemat); break;
        }
      }
    };
    if (!two_phase) {
      // levels of this kind (round 3).  2: box elements with one material each (k_enr_box; the rectilinearity pre-pass and
      // phase 1 of the kernel raise bad[1], the pre-pass also bad[3]); 1: any hexahedra with one material each (k_enr_node
      // with the material outside the Gauss-point loop; phase 1 raises bad[1]); 0: the general form.  On a failure K is
      // assembled again one level down -- or two, when the failed premise is the material's, which level 1 shares -- and
      // the plan stays there
      for (int attempt = 0; attempt < 3; attempt++) {
        const int level = p->fast_level;
        box = level == 2;
        onemat = level == 1;
        if (box) {
          if (int rcr = launch_rect_check(p)) return rcr;
        }
        if (s.ks > 0 && p->send) {
          node_pass(p->sfull, s.ks - 1, s.ks, 1);
          if (int rcp = launch_iface_pack(p)) return rcp;
        }
        node_pass(csr_values, s.ks, s.ke, 0);
        HIPCHK(hipGetLastError());
        if (level == 0) break;
        unsigned long long *fl = p->flags_host;
        fl[1] = fl[2] = fl[3] = 0;
        HIPCHK(hipMemcpyAsync(fl + 1, p->bad + 1, 3 * sizeof(unsigned long long), hipMemcpyDeviceToHost, p->stream));
        HIPCHK(hipStreamSynchronize(p->stream));
        if (!fl[1]) break;
        // (the pre-pass counts a non-box element in bad[1] and in bad[3]: any excess of bad[1] is the material's)
        p->fast_level = (level == 2 && fl[1] == fl[3]) ? 1 : 0;
        HIPCHK(hipMemsetAsync(p->bad, 0, 4 * sizeof(unsigned long long), p->stream));
      }
    }
    for (int64_t L0 = s.els; two_phase && L0 < s.ele; L0 += chunk) {
      const int64_t L1 = (L0 + chunk < s.ele) ? L0 + chunk : s.ele;
      EnrArgs el = a;
      el.L0 = L0;
      el.L1 = L1;
      const int64_t nelem = eplane * (L1 - L0);
      switch (S) {
        case 1: launch_enr_elem<1>(p, el, nelem, false); break;
        case 2: launch_enr_elem<2>(p, el, nelem, mfma); break;
        case 3: launch_enr_elem<3>(p, el, nelem, false); break;
        default: launch_enr_elem<4>(p, el, nelem, false); break;
      }
      HIPCHK(hipGetLastError());
      if (L0 == s.els && s.ks > 0 && p->send) {  // partial rows of node layer ks-1 (element layer els only), owed to the rank below
        gather(p->sfull, s.ks - 1, s.ks, 1);
        if (int rcp = launch_iface_pack(p)) return rcp;
      }
      // node layers this chunk completes: k needs element layers k-1 and k (those this rank assembles)
      const int64_t n0 = (L0 == s.els) ? s.ks : L0, n1 = (L1 == s.ele) ? s.ke : L1;
      gather(csr_values, n0, n1, 0);
      HIPCHK(hipGetLastError());
    }
    if (stats) HIPCHK(hipEventRecord(p->ev[5], p->stream));
  }
  if (stats) HIPCHK(hipEventRecord(p->ev[1], p->stream));
  if (b) {
    if (p->nel > 0) {
      const bool simple = p->opt.enr_rhs_simple != 0;  // the thread-per-element statement of the formulas
      if (simple) {
        const dim3 grid((unsigned)((p->nel + 127) / 128));
        switch (S) {
          case 1: hipLaunchKernelGGL(k_enr_rhs_elem<1>, grid, dim3(128), 0, p->stream, s, a, p->nel); break;
          case 2: hipLaunchKernelGGL(k_enr_rhs_elem<2>, grid, dim3(128), 0, p->stream, s, a, p->nel); break;
          case 3: hipLaunchKernelGGL(k_enr_rhs_elem<3>, grid, dim3(128), 0, p->stream, s, a, p->nel); break;
          default: hipLaunchKernelGGL(k_enr_rhs_elem<4>, grid, dim3(128), 0, p->stream, s, a, p->nel); break;
        }
      } else {
        const dim3 grid((unsigned)((p->nel + 31) / 32));  // 4 waves x 8 elements
        switch (S) {
          case 1: hipLaunchKernelGGL(k_enr_rhs_elem8<1>, grid, dim3(256), 0, p->stream, s, a, p->nel); break;
          case 2: hipLaunchKernelGGL(k_enr_rhs_elem8<2>, grid, dim3(256), 0, p->stream, s, a, p->nel); break;
          case 3: hipLaunchKernelGGL(k_enr_rhs_elem8<3>, grid, dim3(256), 0, p->stream, s, a, p->nel); break;
          default: hipLaunchKernelGGL(k_enr_rhs_elem8<4>, grid, dim3(256), 0, p->stream, s, a, p->nel); break;
        }
      }
    }
    auto rgather = [&](double *out, int64_t l0, int64_t l1, int iface) {
      if (l1 <= l0) return;
      EnrArgs g = a;
      g.out = out;
      g.s0 = l0;
      g.s1 = l1;
      g.iface = iface;
      const int64_t nt = M * N * (l1 - l0) * D;
      hipLaunchKernelGGL(k_enr_rhs_gather, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, p->stream, s, g);
    };
    rgather(b, s.ks, s.ke, 0);
    if (s.ks > 0 && p->send) rgather(p->send + p->send_nnz, s.ks - 1, s.ks, 1);
    HIPCHK(hipGetLastError());
  }
  if (stats) HIPCHK(hipEventRecord(p->ev[2], p->stream));
  unsigned long long *flags = p->flags_host;
  flags[0] = 0;
  HIPCHK(hipMemcpyAsync(flags, p->bad, sizeof(unsigned long long), hipMemcpyDeviceToHost, p->stream));
  HIPCHK(hipStreamSynchronize(p->stream));
  if (stats) {
    float k_ms = 0, r_ms = 0, m_ms = 0;
    HIPCHK(hipEventElapsedTime(&k_ms, p->ev[0], p->ev[1]));
    HIPCHK(hipEventElapsedTime(&r_ms, p->ev[1], p->ev[2]));
    if (csr_values && nlay > 0) HIPCHK(hipEventElapsedTime(&m_ms, p->ev[4], p->ev[5]));
    stats->ms_kernel_K = k_ms;
    stats->ms_kernel_rhs = r_ms;
    stats->ms_kernel_main = m_ms;
    stats->ms_staging = 0.0;
    stats->nel = p->nel;
    stats->nrows = p->nrows;
    stats->nnz = p->nnz;
    stats->bad_detJ = (int64_t)flags[0];
    // algorithmic bytes (SURVEY 8(d) with dof = D): values out, coordinates, E and nu per Gauss point, and per family
    // the 40 doubles of enrichment data the stiffness reads (b: + f, the natural-BC inputs and 24 more per family)
    stats->bytes_K = 8.0 * p->nnz + 8.0 * 3 * p->nnodes + (16.0 * 8 + 8.0 * 40 * p->n_enr) * p->nel;
    stats->bytes_system = stats->bytes_K + 8.0 * p->nrows + (8.0 * 3 * 8 + 8.0 * 264 + 8.0 * 24 * p->n_enr) * p->nel;
    stats->bytes_main = stats->bytes_K;
    stats->ms_total = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
  }
  (void)P;
  if (flags[0]) return fail(FEMASM_ERR_GEOMETRY, "%llu Gauss points with non-positive or NaN det(J)", flags[0]);
  return FEMASM_OK;
}

}  // namespace

extern "C" {

const char *femasm_last_error(void) { return g_err.c_str(); }
const char *femasm_version(void) { return "femasm 0.1 (gfx950)"; }

int femasm_plan_create(const femasm_mesh *mesh, const femasm_quad *quad, int device, femasm_plan **out) {
  if (!mesh || !out) return fail(FEMASM_ERR_ARG, "femasm_plan_create: null argument");
  *out = nullptr;
  int nsd, dof, nen, ngp, nfe;
  const int n_enr = (mesh->problem == KIND_ENRICHED) ? mesh->n_enr : 0;
  if (n_enr < 0 || n_enr > ENR_MAX_S - 1)
    return fail(FEMASM_ERR_ARG, "n_enr %d not in 0..%d (enrichment families per node)", n_enr, ENR_MAX_S - 1);
  if (traits_of(mesh->problem, nsd, dof, nen, ngp, nfe, n_enr))
    return fail(FEMASM_ERR_ARG, "unknown problem kind %d", mesh->problem);
  if (quad && (quad->rule != 0 || (quad->ngp != 0 && quad->ngp != ngp)))
    return fail(FEMASM_ERR_ARG, "only the reference 2-point Gauss rule (rule 0, ngp %d) is supported", ngp);
  const int64_t M = mesh->M, N = mesh->N, P = (nsd == 3) ? mesh->P : 1;
  const int64_t SL = (nsd == 3) ? P : N;  // extent of the slab direction
  if (M < 2 || N < 2 || (nsd == 3 && P < 2)) return fail(FEMASM_ERR_ARG, "need at least 2 nodes per direction");
  if (mesh->ks < 0 || mesh->ke > SL || mesh->ks >= mesh->ke)
    return fail(FEMASM_ERR_ARG, "bad owned layer range [%lld,%lld) of %lld", (long long)mesh->ks, (long long)mesh->ke,
                (long long)SL);
  const int64_t need_lo = mesh->ks > 0 ? mesh->ks - 1 : 0;
  if (mesh->cks > need_lo || mesh->cke < mesh->ke || mesh->cke > SL || mesh->cks < 0)
    return fail(FEMASM_ERR_ARG, "coords layers [%lld,%lld) must cover [%lld,%lld)", (long long)mesh->cks,
                (long long)mesh->cke, (long long)need_lo, (long long)mesh->ke);
  if (!mesh->coords) return fail(FEMASM_ERR_ARG, "coords is null");
  if (N > 65535 || P > 65535) return fail(FEMASM_ERR_ARG, "at most 65535 nodes in the y and z directions (grid limits)");
  if ((double)dof * M * N * P > 2.0e9) return fail(FEMASM_ERR_ARG, "global row count exceeds int32 column ids");

  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(FEMASM_ERR_HIP, "no HIP device: the femasm engine has no CPU path");
  if (device < 0 || device >= ndev) return fail(FEMASM_ERR_ARG, "device %d out of range (%d devices)", device, ndev);
  HIPCHK(hipSetDevice(device));

  femasm_plan *p = new (std::nothrow) femasm_plan();
  if (!p) return fail(FEMASM_ERR_NOMEM, "out of host memory");
  p->device = device;
  p->kind = mesh->problem;
  p->n_enr = n_enr;
  if (int rcs = static_scratch_for(device, &p->static_scratch)) {  // set-up of the no-allocation entry point, once per device
    delete p;
    return rcs;
  }
  Slab &s = p->s;
  s.dim[0] = M; s.dim[1] = N; s.dim[2] = P;
  s.ks = mesh->ks; s.ke = mesh->ke; s.cks = mesh->cks;
  s.nsd = nsd; s.dof = dof;
  // element ownership of the reference (DMDAGetElementsCorners/Sizes): an element belongs to
  // the rank that owns its upper node layer -> layers [ks-1, ke-1), rank 0 starts at 0.
  s.els = mesh->ks > 0 ? mesh->ks - 1 : 0;
  s.ele = mesh->ke - 1;
  s.mat_e_ks = s.els;
  s.before0 = 0;
  s.before0 = (nsd == 3) ? pairs_before(s, 0, 0, s.ks) : pairs_before(s, 0, s.ks, 0);
  p->cke = mesh->cke;

  const int64_t plane = (nsd == 3) ? M * N : M;             // nodes per layer
  const int64_t eplane = (nsd == 3) ? (M - 1) * (N - 1) : (M - 1);
  p->nnodes = plane * (s.ke - s.ks);
  p->nrows = p->nnodes * dof;
  p->nnz = layer_nnz(s, s.ks, s.ke);
  p->nel = eplane * (s.ele - s.els);
  if (p->nel < 0) p->nel = 0;

  int rc = 0;
  auto cleanup = [&]() { femasm_plan_destroy(p); return rc; };
#define PCHK(call)                                                                                   \
  do {                                                                                               \
    hipError_t e_ = (call);                                                                          \
    if (e_ != hipSuccess) {                                                                          \
      rc = fail(FEMASM_ERR_HIP, "%s:%d: %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e_)); \
      return cleanup();                                                                              \
    }                                                                                                \
  } while (0)

  // Default: the legacy null stream, which orders with whatever the caller (PETSc, torch) queued on
  // it or on any blocking stream.  femasm_plan_set_stream() moves the plan to a stream of the caller.
  p->stream = nullptr;
  p->own_stream = false;
  for (auto &e : p->ev) PCHK(hipEventCreate(&e));

  // coordinates
  p->coords_user = mesh->coords;
  p->ncoord = plane * (mesh->cke - mesh->cks) * nsd;
  if (is_device_ptr(mesh->coords)) {
    p->coords = mesh->coords;
  } else {
    PCHK(hipMalloc(&p->coords_dev, p->ncoord * sizeof(double)));
    PCHK(hipMemcpyAsync(p->coords_dev, mesh->coords, p->ncoord * sizeof(double), hipMemcpyHostToDevice, p->stream));
    p->coords = p->coords_dev;
  }

  {  // this plan's slice of the static scratch (set-up of the no-allocation entry point)
    double *base = p->static_scratch;
    p->static_scratch = nullptr;
    p->static_slot = scratch_reserve(device, (p->nel + 255) / 256, &p->static_nslots);
    if (p->static_slot >= 0) p->static_scratch = base + 2 * SCR_SLOT_NB * p->static_slot;
  }
  PCHK(hipMalloc(&p->rowptr, (p->nrows + 1) * sizeof(int64_t)));
  PCHK(hipMalloc(&p->colidx, (size_t)(p->nnz > 0 ? p->nnz : 1) * sizeof(int32_t)));
  {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) p->ncu = prop.multiProcessorCount;
  }
  PCHK(hipHostMalloc(&p->flags_host, 4 * sizeof(unsigned long long), hipHostMallocDefault));
  PCHK(hipMalloc(&p->bad, 32 * sizeof(unsigned long long)));  // [0] bad detJ, [1] fast-path premise failures, [2] hand-shake timeouts, rest: profiling builds
  PCHK(hipMemsetAsync(p->bad, 0, 32 * sizeof(unsigned long long), p->stream));
  // element vectors: blocks of 16 elements x nen*dof lines (ef_idx in femasm_kernels.h)
  p->nelp = ((p->nel > 0 ? p->nel : 1) + 15) & ~(int64_t)15;
  p->elemF_count = p->nelp * nen * dof;
  if (p->kind == KIND_ENRICHED) p->elemF_count = p->nelp * nen * dof;
  if (p->kind == KIND_ELASTIC) {  // or the records of the runs kernel (femasm_kernels.h), whichever is larger
    const int64_t nlay = p->s.ele - p->s.els;
    const int64_t recs = ((nlay + RUNS_ZC - 1) / RUNS_ZC) * (N - 1) * ((M - 1 + 15) / 16) * (RUNS_ZC + 1) * 128;
    if (recs > p->elemF_count) p->elemF_count = recs;
  }
  PCHK(hipMalloc(&p->elemF, (size_t)p->elemF_count * sizeof(double)));

  // interface buffers: rows of node layer ks-1 (owed to the rank below) and of ke-1 (owed to us)
  const int64_t packed_nnz = (int64_t)dof * dof * 2 * ((nsd == 3) ? tot1(N) * tot1(M) : tot1(M));
  if (s.ks > 0) {
    p->sfull_nnz = layer_nnz(s, s.ks - 1, s.ks);
    PCHK(hipMalloc(&p->sfull, (size_t)p->sfull_nnz * sizeof(double)));
    p->send_nnz = packed_nnz;
    p->send_count = p->send_nnz + plane * dof;
    PCHK(hipMalloc(&p->send, p->send_count * sizeof(double)));
  }
  if (s.ke < SL) {
    p->recv_nnz = packed_nnz;
    p->recv_count = p->recv_nnz + plane * dof;
    PCHK(hipMalloc(&p->recv, p->recv_count * sizeof(double)));
  }

  hipLaunchKernelGGL(k_pattern, dim3(nblk(p->nrows + 1)), dim3(256), 0, p->stream, s, p->nrows, p->rowptr, p->colidx);
  PCHK(hipGetLastError());
  PCHK(hipStreamSynchronize(p->stream));
#undef PCHK
  *out = p;
  return FEMASM_OK;
}

int femasm_plan_destroy(femasm_plan *p) {
  if (!p) return FEMASM_OK;
  (void)hipSetDevice(p->device);
  if (p->stream) (void)hipStreamSynchronize(p->stream);
#if PH_EXP == 8
  {  // wave placement of k_assemble_p3_hex4 (profiling build)
    unsigned long long t[32];
    if (p->bad && hipMemcpy(t, p->bad, sizeof t, hipMemcpyDeviceToHost) == hipSuccess && t[8] + t[9] + t[10] + t[11]) {
      for (int w = 0; w < 8; w++)  // (SIMD + 1 of wave w in four sample workgroups, one byte each, summed over the launches)
        fprintf(stderr, "[femasm p3hex4 placement] wave %d: %llx\n", w, t[8 + w]);
      fprintf(stderr, "[femasm p3hex4 placement] element waves on SIMD 0..3: %llu %llu %llu %llu\n", t[24], t[25], t[26], t[27]);
    }
  }
#endif
#if PH_EXP == 9
  {  // phase clocks of k_assemble_p3_hex (profiling build): summed over all waves and launches, per wave-step
    unsigned long long t[32];
    if (p->bad && hipMemcpy(t, p->bad, sizeof t, hipMemcpyDeviceToHost) == hipSuccess && t[16]) {
      const char *nm[8] = {"coords", "B1", "element", "B2", "gather", "B3", "stage+B4", "store"};
      fprintf(stderr, "[femasm p3hex phases] cycles per wave-step (%llu wave-steps):", t[16]);
      for (int q = 0; q < 8; q++) fprintf(stderr, " %s %.0f |", nm[q], (double)t[8 + q] / t[16]);
      fprintf(stderr, "\n");
    }
  }
#endif
#if H2_EXP == 9
  {  // phase clocks of k_assemble_s3_hex2 (profiling build): one workgroup, two of its waves, summed over launches
    unsigned long long t[32];
    if (p->bad && hipMemcpy(t, p->bad, sizeof t, hipMemcpyDeviceToHost) == hipSuccess && t[24]) {
      const char *nm[8] = {"record", "B1", "contract", "B2", "deposit", "B3", "store", "B4"};
      fprintf(stderr, "[femasm hex phases] most workgroups seen together on one CU: %llu; cycles per step (steps %llu):", t[25], t[24]);
      for (int q = 0; q < 8; q++) fprintf(stderr, " %s w0 %.0f w8 %.0f |", nm[q], (double)t[8 + q] / t[24], (double)t[16 + q] / t[24]);
      fprintf(stderr, "\n");
      if (t[27]) fprintf(stderr, "[femasm hex phases] workgroups %llu, mean life %.1f us = %.0f cycles (%.2f GHz)\n", t[27], (double)t[26] / t[27] / 100.0, (double)t[28] / t[27], (double)t[28] / t[26] / 10.0);
    }
  }
#endif
#if FEMASM_EXP == 3
  {  // phase clocks of the row-sweep kernel (profiling build)
    unsigned long long t[32];
    if (p->bad && hipMemcpy(t, p->bad, sizeof t, hipMemcpyDeviceToHost) == hipSuccess) {
      fprintf(stderr, "[femasm phases] row waves (s3 unit kernel, sampled waves: units, loads+record, rows, stage reads, stores): n=%llu barrier %llu contract %llu stage %llu store %llu | record wave: n=%llu top %llu commit+load %llu build %llu handshakes %llu\n",
              t[8], t[9], t[10], t[11], t[12], t[16], t[17], t[18], t[19], t[20]);
      fprintf(stderr, "[femasm phases] rhs coop: n=%llu commit %llu request %llu phaseA %llu phaseB %llu\n", t[24], t[25], t[26],
              t[27], t[28]);
    }
  }
#endif
  void *bufs[] = {p->coords_dev, p->rowptr, p->colidx, p->elemF, p->own_send ? p->send : nullptr,
                  p->own_recv ? p->recv : nullptr, p->bad, p->stage_vals, p->stage_b, p->stage_gp, p->lines, p->partial, p->stage_u, p->ke_ring, p->sfull};
  for (void *b : bufs)
    if (b) (void)hipFree(b);
  if (p->flags_host) (void)hipHostFree(p->flags_host);
  for (auto &e : p->ev)
    if (e) (void)hipEventDestroy(e);
  scratch_release(p->device, p->static_slot, p->static_nslots);
  if (p->comm_stream) {
    (void)hipStreamSynchronize(p->comm_stream);
    (void)hipStreamDestroy(p->comm_stream);
  }
  if (p->ev_send) (void)hipEventDestroy(p->ev_send);
  if (p->ev_xchg) (void)hipEventDestroy(p->ev_xchg);
  if (p->own_stream && p->stream) (void)hipStreamDestroy(p->stream);
  delete p;
  return FEMASM_OK;
}

int femasm_plan_set_stream(femasm_plan *p, void *stream) {
  if (!p) return fail(FEMASM_ERR_ARG, "null plan");
  HIPCHK(hipSetDevice(p->device));
  HIPCHK(hipStreamSynchronize(p->stream));
  p->stream = (hipStream_t)stream;
  p->own_stream = false;
  return FEMASM_OK;
}

int femasm_plan_set_option(femasm_plan *p, const char *name, int64_t value) {
  if (!p || !name) return fail(FEMASM_ERR_ARG, "null argument");
  if (value < 0) return fail(FEMASM_ERR_ARG, "option %s: value %lld is negative (0 = automatic)", name, (long long)value);
  const std::string n(name);
  auto &o = p->opt;
  if (n == "zc") o.zc = value;
  else if (n == "s3_units") o.s3_units = value;
  else if (n == "e2_yc") o.e2_yc = value;
  else if (n == "rhs_grid") o.rhs_grid = value;
  else if (n == "enr_scratch_mb") o.enr_scratch_mb = value;
  else if (n == "s3_rowsweep") o.s3_rowsweep = value != 0;
  else if (n == "p3_sweep") o.p3_sweep = (value == 2) ? 2 : (value != 0);
  else if (n == "e2_brick") o.e2_brick = value != 0;
  else if (n == "e2_lane_loads") o.e2_lane_loads = value != 0;
  else if (n == "e2_two_pass") o.e2_two_pass = value != 0;
  else if (n == "p3_hex2w") o.p3_hex2w = value != 0;
  else if (n == "rhs_simple") o.rhs_simple = value != 0;
  else if (n == "enr_rhs_simple") o.enr_rhs_simple = value != 0;
  else if (n == "debug") o.debug = value != 0;
  else return fail(FEMASM_ERR_ARG, "unknown option '%s'", name);
  return FEMASM_OK;
}

int femasm_plan_sync(femasm_plan *p) {
  if (!p) return fail(FEMASM_ERR_ARG, "null plan");
  HIPCHK(hipSetDevice(p->device));
  if (p->pending) return finish_pending(p);
  HIPCHK(hipStreamSynchronize(p->stream));
  return FEMASM_OK;
}

int femasm_plan_set_async(femasm_plan *p, int on) {
  if (!p) return fail(FEMASM_ERR_ARG, "null plan");
  HIPCHK(hipSetDevice(p->device));
  if (int rc = finish_pending(p)) return rc;
  p->async = on != 0;
  return FEMASM_OK;
}

int femasm_pattern(femasm_plan *p, int64_t *nrows, int64_t *nnz, const int64_t **rowptr, const int32_t **colidx) {
  if (!p) return fail(FEMASM_ERR_ARG, "null plan");
  if (nrows) *nrows = p->nrows;
  if (nnz) *nnz = p->nnz;
  if (rowptr) *rowptr = p->rowptr;
  if (colidx) *colidx = p->colidx;
  return FEMASM_OK;
}

int femasm_pattern_copy(femasm_plan *p, int64_t *rowptr, int32_t *colidx) {
  if (!p) return fail(FEMASM_ERR_ARG, "null plan");
  HIPCHK(hipSetDevice(p->device));
  if (rowptr) HIPCHK(hipMemcpy(rowptr, p->rowptr, (p->nrows + 1) * sizeof(int64_t), hipMemcpyDeviceToHost));
  if (colidx) HIPCHK(hipMemcpy(colidx, p->colidx, p->nnz * sizeof(int32_t), hipMemcpyDeviceToHost));
  return FEMASM_OK;
}

int femasm_scatter_map(femasm_plan *p, int64_t *map_out) {
  if (!p || !map_out) return fail(FEMASM_ERR_ARG, "null argument");
  HIPCHK(hipSetDevice(p->device));
  int nsd, dof, nen, ngp, nfe;
  traits_of(p->kind, nsd, dof, nen, ngp, nfe, p->n_enr);
  const int nr = nen * dof;
  const int64_t total = p->nel * nr * nr;
  if (total == 0) return FEMASM_OK;
  int64_t *dst = map_out;
  const bool dev = is_device_ptr(map_out);
  if (!dev) HIPCHK(hipMalloc(&dst, total * sizeof(int64_t)));
  hipLaunchKernelGGL(k_scatter_map, dim3(nblk(p->nel * nr)), dim3(256), 0, p->stream, p->s, p->nel, nen, dst);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipStreamSynchronize(p->stream);
  if (e == hipSuccess && !dev) e = hipMemcpy(map_out, dst, total * sizeof(int64_t), hipMemcpyDeviceToHost);
  if (!dev) (void)hipFree(dst);
  if (e != hipSuccess) return fail(FEMASM_ERR_HIP, "femasm_scatter_map: %s", hipGetErrorString(e));
  return FEMASM_OK;
}

int femasm_assemble_system(femasm_plan *p, const femasm_material *mat, const femasm_bc *bcs, double *csr_values,
                           double *b, femasm_stats *stats) {
  if (!p) return fail(FEMASM_ERR_ARG, "null plan");
  const bool iface_done = p->iface_done;  // consumed by this call, however it ends
  p->iface_done = false;
  HIPCHK(hipSetDevice(p->device));
  if (int rcp = finish_pending(p)) return rcp;  // flags of an asynchronous call before this one: its errors surface here
  const auto t_begin = std::chrono::steady_clock::now();
  const int bc = bcs ? bcs->kind : FEMASM_BC_NONE;
  const Slab &s = p->s;
  int nsd, dof, nen, ngp, nfe;
  traits_of(p->kind, nsd, dof, nen, ngp, nfe, p->n_enr);

  // argument checks, before anything is launched
  if (bc < 0 || bc > 3) return fail(FEMASM_ERR_ARG, "unknown bc kind %d", bc);
  const bool symmetric = bcs && bcs->symmetric != 0 && bc != FEMASM_BC_NONE;
  if (symmetric) {
    const int64_t SLs = (s.nsd == 3) ? s.dim[2] : s.dim[1];
    if (p->kind == KIND_ENRICHED) return fail(FEMASM_ERR_ARG, "symmetric elimination is not offered for the enriched kind");
    if (s.ks != 0 || s.ke != SLs)
      return fail(FEMASM_ERR_ARG, "on a slab, call femasm_eliminate_columns after femasm_interface_apply (the rows must be complete)");
    if (!csr_values || !is_device_ptr(csr_values) || (b && !is_device_ptr(b)))
      return fail(FEMASM_ERR_ARG, "symmetric elimination works on device-resident csr_values / b");
  }
  if (p->kind == KIND_ENRICHED) return assemble_enriched(p, mat, bc, csr_values, b, stats);
  if ((bc == FEMASM_BC_POISSON_INLOOP && p->kind != KIND_POISSON) ||
      (bc == FEMASM_BC_E2_COMPRESSION && p->kind != KIND_PLANE) ||
      (bc == FEMASM_BC_S3_ALLFACES_EXACT && p->kind != KIND_ELASTIC))
    return fail(FEMASM_ERR_ARG, "bc kind %d does not belong to problem kind %d", bc, p->kind);
  femasm_material mloc{};
  if (p->kind != KIND_POISSON) {
    if (!mat) return fail(FEMASM_ERR_ARG, "material is required for elastic problems");
    mloc = *mat;
    if (mloc.layout == FEMASM_MAT_AOS) {
      const int need = (p->kind == KIND_PLANE) ? 24 : 400;
      if (!mloc.gp) return fail(FEMASM_ERR_ARG, "material.gp is null");
      if (mloc.stride < need) return fail(FEMASM_ERR_ARG, "material stride %d < %d", mloc.stride, need);
      if (mloc.e_ks > s.els) return fail(FEMASM_ERR_ARG, "material starts at element layer %lld > %lld",
                                         (long long)mloc.e_ks, (long long)s.els);
    } else if (mloc.layout == FEMASM_MAT_ELEMENT_SOA) {
      if (const char *why = check_element_soa(p, mloc)) return fail(FEMASM_ERR_ARG, "%s", why);
    } else if (mloc.layout != FEMASM_MAT_UNIFORM && !(mloc.layout == FEMASM_MAT_MANUFACTURED && p->kind == KIND_ELASTIC)) {
      return fail(FEMASM_ERR_ARG, "unknown material layout %d", mloc.layout);
    }
    p->s.mat_e_ks = (mloc.layout == FEMASM_MAT_AOS || mloc.layout == FEMASM_MAT_ELEMENT_SOA) ? mloc.e_ks : s.els;
  } else {
    mloc.layout = FEMASM_MAT_UNIFORM;
  }

  // host-side callers: stage inputs/outputs through device buffers
  double ms_staging = 0.0;
  const bool vals_host = csr_values && !is_device_ptr(csr_values);
  const bool b_host = b && !is_device_ptr(b);
  double *d_vals = csr_values, *d_b = b;
  auto t0 = std::chrono::steady_clock::now();
  if (vals_host) {
    if (!p->stage_vals) HIPCHK(hipMalloc(&p->stage_vals, (size_t)(p->nnz > 0 ? p->nnz : 1) * sizeof(double)));
    d_vals = p->stage_vals;
  }
  if (b_host) {
    if (!p->stage_b) HIPCHK(hipMalloc(&p->stage_b, (size_t)p->nrows * sizeof(double)));
    d_b = p->stage_b;
  }
  p->gp_cur = ((mloc.layout == FEMASM_MAT_AOS || mloc.layout == FEMASM_MAT_ELEMENT_SOA) && p->kind != KIND_POISSON) ? mloc.gp : nullptr;
  if (p->gp_cur && !is_device_ptr(mloc.gp)) {
    const int64_t eplane = (nsd == 3) ? (s.dim[0] - 1) * (s.dim[1] - 1) : (s.dim[0] - 1);
    const int64_t cnt = eplane * (s.ele - mloc.e_ks) * mloc.stride;
    if (cnt > 0) {
      if (cnt > p->stage_gp_count) {  // plan-owned, reallocated only when the required size grows
        if (p->stage_gp) (void)hipFree(p->stage_gp);
        p->stage_gp = nullptr;
        p->stage_gp_count = 0;
        HIPCHK(hipMalloc(&p->stage_gp, cnt * sizeof(double)));
        p->stage_gp_count = cnt;
      }
      HIPCHK(hipMemcpyAsync(p->stage_gp, mloc.gp, cnt * sizeof(double), hipMemcpyHostToDevice, p->stream));
      HIPCHK(hipStreamSynchronize(p->stream));
      p->gp_cur = p->stage_gp;
    }
  }
  ms_staging += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();

  int rc = 0;
  if (stats) HIPCHK(hipEventRecord(p->ev[0], p->stream));  // ms_kernel_K includes the rectilinearity pre-pass
  if (iface_done) {
    // femasm_assemble_interface ran just before on the same inputs without reading its flags back: they carry
    // over (a broken premise or a timed-out hand-shake there is found here), and so does the verdict of its
    // rectilinearity check; the count of bad Gauss points starts again (this call integrates those elements too)
    HIPCHK(hipMemsetAsync(p->bad, 0, sizeof(unsigned long long), p->stream));
    p->rect_checked = p->iface_rect && p->fast_level == 2;
  } else {
    HIPCHK(hipMemsetAsync(p->bad, 0, 4 * sizeof(unsigned long long), p->stream));
    rc = launch_rect_check(p);
    if (rc) return rc;
  }
  // ---- K ----
  auto run_K = [&]() -> int {
    int r = 0;
    switch (p->kind) {
      case KIND_POISSON: r = launch_K<KIND_POISSON>(p, &mloc, bc, d_vals, s.ks, s.ke, 0); break;
      case KIND_PLANE:   r = launch_K<KIND_PLANE>(p, &mloc, bc, d_vals, s.ks, s.ke, 0); break;
      case KIND_ELASTIC: r = launch_K<KIND_ELASTIC>(p, &mloc, bc, d_vals, s.ks, s.ke, 0); break;
    }
    if (r) return r;
    if (s.ks > 0 && !iface_done) {  // partial rows of node layer ks-1, owed to the rank below
      switch (p->kind) {
        case KIND_POISSON: r = launch_K<KIND_POISSON>(p, &mloc, bc, p->sfull, s.ks - 1, s.ks, 1); break;
        case KIND_PLANE:   r = launch_K<KIND_PLANE>(p, &mloc, bc, p->sfull, s.ks - 1, s.ks, 1); break;
        case KIND_ELASTIC: r = launch_K<KIND_ELASTIC>(p, &mloc, bc, p->sfull, s.ks - 1, s.ks, 1); break;
      }
      if (!r) r = launch_iface_pack(p);
    }
    return r;
  };
  p->used_fast = false;
  p->time_main = stats != nullptr;
  p->bytes_main = 0.0;
  p->e2_bout = (p->kind == KIND_PLANE && d_vals && d_b) ? d_b : nullptr;
  p->e2_b0 = p->e2_b1 = 0;
  if (d_vals) {
    rc = run_K();
    if (rc) return rc;
  }
  p->e2_bout = nullptr;  // (the conditional and fallback passes below assemble K only)
  if (stats) HIPCHK(hipEventRecord(p->ev[1], p->stream));
  // ---- b ----
  if (d_b) {
    switch (p->kind) {
      case KIND_POISSON: rc = launch_rhs<KIND_POISSON>(p, &mloc, bc, d_b, s.ks, s.ke, 0, true); break;
      case KIND_PLANE:
        if (p->e2_b1 > p->e2_b0) {  // the stiffness kernel wrote b of the node rows [e2_b0, e2_b1): the rows around them
          rc = launch_rhs<KIND_PLANE>(p, &mloc, bc, d_b, s.ks, p->e2_b0, 0, true);
          if (!rc) rc = launch_rhs<KIND_PLANE>(p, &mloc, bc, d_b + 2 * s.dim[0] * (p->e2_b1 - s.ks), p->e2_b1, s.ke, 0, true);
        } else {
          rc = launch_rhs<KIND_PLANE>(p, &mloc, bc, d_b, s.ks, s.ke, 0, true);
        }
        break;
      case KIND_ELASTIC: rc = launch_rhs<KIND_ELASTIC>(p, &mloc, bc, d_b, s.ks, s.ke, 0, true); break;
    }
    if (rc) return rc;
    if (s.ks > 0 && !iface_done) {
      double *sb = p->send + p->send_nnz;
      switch (p->kind) {
        case KIND_POISSON: rc = launch_rhs<KIND_POISSON>(p, &mloc, bc, sb, s.ks - 1, s.ks, 1, false); break;
        case KIND_PLANE:   rc = launch_rhs<KIND_PLANE>(p, &mloc, bc, sb, s.ks - 1, s.ks, 1, false); break;
        case KIND_ELASTIC: rc = launch_rhs<KIND_ELASTIC>(p, &mloc, bc, sb, s.ks - 1, s.ks, 1, false); break;
      }
      if (rc) return rc;
    }
  }
  if (stats) HIPCHK(hipEventRecord(p->ev[2], p->stream));

  if (p->async && !stats && !vals_host && !b_host && !(p->gp_cur && p->gp_cur == p->stage_gp)) {
    // asynchronous form: nothing is read back here.  K has to be final in stream order whatever the mesh, so the general
    // kernels are queued behind the fast ones on the same rows and run only if the premise flag came up on the device
    // (the scheme of femasm_assemble_interface); flags and the plan's level are dealt with by femasm_plan_sync or the next call.
    if (d_vals && p->used_fast) {
      const int level = p->fast_level;
      p->fast_level = 0;
      p->cond_general = true;
      p->time_main = false;
      rc = run_K();
      p->cond_general = false;
      p->fast_level = level;
      if (rc) return rc;
      p->pending_fast = true;
      if (p->e2_b1 > p->e2_b0) {  // b of the one-pass kernel's rows again, on the device's verdict
        p->cond_rhs = true;
        rc = launch_rhs<KIND_PLANE>(p, &mloc, bc, d_b + 2 * s.dim[0] * (p->e2_b0 - s.ks), p->e2_b0, p->e2_b1, 0, true);
        p->cond_rhs = false;
        if (rc) return rc;
      }
    }
    p->pending = true;
    if (symmetric) return femasm_eliminate_columns(p, bcs, d_vals, d_b);
    return FEMASM_OK;
  }
  t0 = std::chrono::steady_clock::now();
  if (vals_host) HIPCHK(hipMemcpyAsync(csr_values, d_vals, p->nnz * sizeof(double), hipMemcpyDeviceToHost, p->stream));
  if (b_host) HIPCHK(hipMemcpyAsync(b, d_b, p->nrows * sizeof(double), hipMemcpyDeviceToHost, p->stream));
  unsigned long long *flags = p->flags_host;
  flags[0] = flags[1] = flags[2] = 0;
  HIPCHK(hipMemcpyAsync(flags, p->bad, 3 * sizeof(unsigned long long), hipMemcpyDeviceToHost, p->stream));
  HIPCHK(hipStreamSynchronize(p->stream));
  if (vals_host || b_host)
    ms_staging += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();

  double k_ms_redo = -1.0;
  while (d_vals && p->used_fast && flags[1]) {
    // The fast kernels verify their premise exactly (level 2: rectilinear mesh, one (E,nu) per
    // element; level 1: elastic, one (E,nu) per element; Poisson, only when the caller set it: parallelepipeds).  The
    // test failed somewhere: the plan drops for good (elastic 2 -> 1 -> 0; Poisson 2 -> 0, 1 -> 0: since round 3 its
    // general kernel is faster than the parallelepiped kernel, 1.26 against 1.51 ms at 256^3) and K is assembled again,
    // overwriting every value of the previous attempt.
    p->fast_level = (p->kind == KIND_ELASTIC && p->fast_level == 2) ? 1 : 0;
    p->used_fast = false;
    HIPCHK(hipMemsetAsync(p->bad, 0, 3 * sizeof(unsigned long long), p->stream));
    rc = launch_rect_check(p);  // (only if the plan is still at level 2, which it is not)
    if (rc) return rc;
    t0 = std::chrono::steady_clock::now();
    rc = run_K();
    if (rc) return rc;
    if (p->e2_b1 > p->e2_b0) {  // the one-pass kernel's premise failed: its b rows again, by the general kernel
      rc = launch_rhs<KIND_PLANE>(p, &mloc, bc, d_b + 2 * s.dim[0] * (p->e2_b0 - s.ks), p->e2_b0, p->e2_b1, 0, true);
      if (rc) return rc;
      p->e2_b0 = p->e2_b1 = 0;
      if (b_host) HIPCHK(hipMemcpyAsync(b, d_b, p->nrows * sizeof(double), hipMemcpyDeviceToHost, p->stream));
    }
    HIPCHK(hipMemcpyAsync(flags, p->bad, 3 * sizeof(unsigned long long), hipMemcpyDeviceToHost, p->stream));
    HIPCHK(hipStreamSynchronize(p->stream));
    k_ms_redo = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (vals_host && !(p->used_fast && flags[1])) {
      t0 = std::chrono::steady_clock::now();
      HIPCHK(hipMemcpyAsync(csr_values, d_vals, p->nnz * sizeof(double), hipMemcpyDeviceToHost, p->stream));
      HIPCHK(hipStreamSynchronize(p->stream));
      ms_staging += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
  }
  p->iface_done = false;
  if (flags[2]) return fail(FEMASM_ERR_HIP, "internal: %llu wave hand-shakes of the row-sweep kernel timed out", flags[2]);
  const unsigned long long bad = flags[0];

  if (stats) {
    float k_ms = 0, r_ms = 0;
    HIPCHK(hipEventElapsedTime(&k_ms, p->ev[0], p->ev[1]));
    HIPCHK(hipEventElapsedTime(&r_ms, p->ev[1], p->ev[2]));
    stats->ms_kernel_K = k_ms_redo >= 0.0 ? k_ms_redo : k_ms;
    float m_ms = 0;
    if (d_vals && p->bytes_main > 0.0) HIPCHK(hipEventElapsedTime(&m_ms, p->ev[4], p->ev[5]));
    stats->ms_kernel_main = m_ms;
    stats->bytes_main = p->bytes_main;
    stats->ms_kernel_rhs = r_ms;
    stats->ms_staging = ms_staging;
    stats->nel = p->nel;
    stats->nrows = p->nrows;
    stats->nnz = p->nnz;
    stats->bad_detJ = (int64_t)bad;
    const bool has_mat = p->kind != KIND_POISSON;
    const bool aos = has_mat && mloc.layout == FEMASM_MAT_AOS;
    stats->bytes_K = 8.0 * p->nnz + 8.0 * nsd * p->nnodes + (has_mat ? 16.0 * ngp * p->nel : 0.0);
    stats->bytes_system = stats->bytes_K + 8.0 * p->nrows + (aos ? 8.0 * dof * ngp * p->nel : 0.0) +
                          ((aos && p->kind == KIND_ELASTIC) ? 8.0 * 264 * p->nel : 0.0);
    stats->ms_total = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
  }
  p->iface_done = false;
  if (bad) return fail(FEMASM_ERR_GEOMETRY, "%llu Gauss points with non-positive or NaN det(J)", bad);
  if (symmetric) return femasm_eliminate_columns(p, bcs, d_vals, d_b);
  return FEMASM_OK;
}

int femasm_plan_set_fast_path(femasm_plan *p, int level) {
  if (!p) return fail(FEMASM_ERR_ARG, "null plan");
  if (level < 0 || level > 2) return fail(FEMASM_ERR_ARG, "fast-path level %d not in 0..2", level);
  p->fast_level = level;
  return FEMASM_OK;
}

int femasm_plan_set_enriched_contraction(femasm_plan *p, int mode) {
  if (!p) return fail(FEMASM_ERR_ARG, "null plan");
  if (p->kind != KIND_ENRICHED) return fail(FEMASM_ERR_ARG, "not an enriched plan");
  if (mode < -1 || mode > 2)
    return fail(FEMASM_ERR_ARG, "mode %d: -1 default (= 2), 0 element matrices on VALU, 1 on FP64 MFMA, 2 node-centric pass", mode);
  if (mode == 1 && p->n_enr != 1)
    return fail(FEMASM_ERR_ARG, "the FP64 MFMA contraction exists for n_enr = 1 (16 scalar functions = one 16x16x4 tile)");
  p->enr_mfma = mode;
  return FEMASM_OK;
}

int femasm_plan_get_fast_path(femasm_plan *p, int *level) {
  if (!p || !level) return fail(FEMASM_ERR_ARG, "null argument");
  *level = p->fast_level;
  return FEMASM_OK;
}

int femasm_assemble_interface(femasm_plan *p, const femasm_material *mat, const femasm_bc *bcs) {
  if (!p) return fail(FEMASM_ERR_ARG, "null plan");
  if (p->kind == KIND_ENRICHED)
    return fail(FEMASM_ERR_ARG, "the enriched kind has no split call: femasm_assemble_system fills the send buffer");
  HIPCHK(hipSetDevice(p->device));
  if (int rcp = finish_pending(p)) return rcp;
  HIPCHK(hipSetDevice(p->device));
  const int bc = bcs ? bcs->kind : FEMASM_BC_NONE;
  if (bc < 0 || bc > 3) return fail(FEMASM_ERR_ARG, "unknown bc kind %d", bc);
  if ((bc == FEMASM_BC_POISSON_INLOOP && p->kind != KIND_POISSON) ||
      (bc == FEMASM_BC_E2_COMPRESSION && p->kind != KIND_PLANE) ||
      (bc == FEMASM_BC_S3_ALLFACES_EXACT && p->kind != KIND_ELASTIC))
    return fail(FEMASM_ERR_ARG, "bc kind %d does not belong to problem kind %d", bc, p->kind);
  femasm_material mloc{};
  if (p->kind != KIND_POISSON) {
    if (!mat) return fail(FEMASM_ERR_ARG, "material is required for elastic problems");
    mloc = *mat;
    if (mloc.layout == FEMASM_MAT_AOS) {
      const int need = (p->kind == KIND_PLANE) ? 24 : 400;
      if (!mloc.gp || !is_device_ptr(mloc.gp))
        return fail(FEMASM_ERR_ARG, "femasm_assemble_interface needs the material array in device memory");
      if (mloc.stride < need) return fail(FEMASM_ERR_ARG, "material stride %d < %d", mloc.stride, need);
      if (mloc.e_ks > p->s.els) return fail(FEMASM_ERR_ARG, "material starts above the first assembled element layer");
    } else if (mloc.layout == FEMASM_MAT_ELEMENT_SOA) {
      if (const char *why = check_element_soa(p, mloc)) return fail(FEMASM_ERR_ARG, "%s", why);
    } else if (mloc.layout != FEMASM_MAT_UNIFORM && !(mloc.layout == FEMASM_MAT_MANUFACTURED && p->kind == KIND_ELASTIC)) {
      return fail(FEMASM_ERR_ARG, "unknown material layout %d", mloc.layout);
    }
    p->s.mat_e_ks = (mloc.layout == FEMASM_MAT_AOS || mloc.layout == FEMASM_MAT_ELEMENT_SOA) ? mloc.e_ks : p->s.els;
  } else {
    mloc.layout = FEMASM_MAT_UNIFORM;
  }
  p->gp_cur = ((mloc.layout == FEMASM_MAT_AOS || mloc.layout == FEMASM_MAT_ELEMENT_SOA) && p->kind != KIND_POISSON) ? mloc.gp : nullptr;  // device memory (checked above)
  const Slab &s = p->s;
  if (s.ks == 0) { p->iface_done = false; return FEMASM_OK; }  // lowest slab: nothing is owed, nothing carries over
  if (!p->send) return fail(FEMASM_ERR_ARG, "no send buffer");
  p->time_main = false;
  int rc = 0;
  auto layer_K = [&]() -> int {
    switch (p->kind) {
      case KIND_POISSON: return launch_K<KIND_POISSON>(p, &mloc, bc, p->sfull, s.ks - 1, s.ks, 1);
      case KIND_PLANE:   return launch_K<KIND_PLANE>(p, &mloc, bc, p->sfull, s.ks - 1, s.ks, 1);
      default:           return launch_K<KIND_ELASTIC>(p, &mloc, bc, p->sfull, s.ks - 1, s.ks, 1);
    }
  };
  p->used_fast = false;
  HIPCHK(hipMemsetAsync(p->bad, 0, 4 * sizeof(unsigned long long), p->stream));
  rc = launch_rect_check(p);
  if (rc) return rc;
  p->iface_rect = p->rect_checked;
  rc = layer_K();
  if (rc) return rc;
  if (p->used_fast) {
    // The caller is about to hand the buffer to the network, so it has to be final when the stream gets there, and
    // the host does not wait for the fast kernel's verdict: the general kernels are queued behind it on the same
    // rows and return at once unless the premise flag is set (one layer: their full run is 1/layers of a general
    // assembly).  Flags are read back by the femasm_assemble_system call that follows, which also drops the
    // plan's level.
    const int level = p->fast_level;
    p->fast_level = 0;
    p->cond_general = true;
    rc = layer_K();
    p->cond_general = false;
    p->fast_level = level;
    p->used_fast = false;
    if (rc) return rc;
  }
  rc = launch_iface_pack(p);  // (behind the conditional general kernels: the rows are final here)
  if (rc) return rc;
  // element vectors of the one element layer that touches node layer ks-1 from above (it is the first layer
  // of the plan's element-vector array, where the full call will write the same numbers again)
  Slab sl = s;
  sl.ele = sl.els + 1;
  const int64_t eplane = (s.nsd == 3) ? (s.dim[0] - 1) * (s.dim[1] - 1) : (s.dim[0] - 1);
  double *sb = p->send + p->send_nnz;
  switch (p->kind) {
    case KIND_POISSON: rc = launch_rhs<KIND_POISSON>(p, &mloc, bc, sb, s.ks - 1, s.ks, 1, true, &sl, eplane); break;
    case KIND_PLANE:   rc = launch_rhs<KIND_PLANE>(p, &mloc, bc, sb, s.ks - 1, s.ks, 1, true, &sl, eplane); break;
    case KIND_ELASTIC: rc = launch_rhs<KIND_ELASTIC>(p, &mloc, bc, sb, s.ks - 1, s.ks, 1, true, &sl, eplane); break;
  }
  if (rc) return rc;
  p->iface_done = true;
  return FEMASM_OK;
}

int femasm_apply_operator(femasm_plan *p, const femasm_material *mat, const femasm_bc *bcs, const double *x,
                          double *y) {
  if (!p || !x || !y) return fail(FEMASM_ERR_ARG, "null argument");
  if (p->kind == KIND_ENRICHED) return fail(FEMASM_ERR_ARG, "femasm_apply_operator: not offered for the enriched kind");
  HIPCHK(hipSetDevice(p->device));
  const int bc = bcs ? bcs->kind : FEMASM_BC_NONE;
  if (bc < 0 || bc > 3) return fail(FEMASM_ERR_ARG, "unknown bc kind %d", bc);
  if ((bc == FEMASM_BC_POISSON_INLOOP && p->kind != KIND_POISSON) ||
      (bc == FEMASM_BC_E2_COMPRESSION && p->kind != KIND_PLANE) ||
      (bc == FEMASM_BC_S3_ALLFACES_EXACT && p->kind != KIND_ELASTIC))
    return fail(FEMASM_ERR_ARG, "bc kind %d does not belong to problem kind %d", bc, p->kind);
  femasm_material mloc{};
  mloc.layout = FEMASM_MAT_UNIFORM;
  if (p->kind != KIND_POISSON) {
    if (!mat) return fail(FEMASM_ERR_ARG, "material is required for elastic problems");
    mloc = *mat;
    if (mloc.layout == FEMASM_MAT_AOS) {
      const int need = (p->kind == KIND_PLANE) ? 24 : 400;
      if (!mloc.gp || !is_device_ptr(mloc.gp) || mloc.stride < need || mloc.e_ks > p->s.els)
        return fail(FEMASM_ERR_ARG, "femasm_apply_operator needs a device material array covering the slab's elements");
    } else if (mloc.layout == FEMASM_MAT_ELEMENT_SOA) {
      if (const char *why = check_element_soa(p, mloc)) return fail(FEMASM_ERR_ARG, "%s", why);
    } else if (mloc.layout != FEMASM_MAT_UNIFORM && !(mloc.layout == FEMASM_MAT_MANUFACTURED && p->kind == KIND_ELASTIC)) {
      return fail(FEMASM_ERR_ARG, "unknown material layout %d", mloc.layout);
    }
    p->s.mat_e_ks = (mloc.layout == FEMASM_MAT_AOS || mloc.layout == FEMASM_MAT_ELEMENT_SOA) ? mloc.e_ks : p->s.els;
  }
  if (!is_device_ptr(x) || !is_device_ptr(y)) return fail(FEMASM_ERR_ARG, "femasm_apply_operator works on device vectors");
  const Slab &s = p->s;
  RArgs a{};
  a.coords = p->coords;
  p->gp_cur = (mloc.layout == FEMASM_MAT_AOS || mloc.layout == FEMASM_MAT_ELEMENT_SOA) ? mloc.gp : nullptr;
  a.gp = p->gp_cur;
  a.elemF = p->elemF;
  a.cke = p->cke;
  a.nel = p->nel;
  a.mat_layout = mloc.layout;
  a.mat_stride = mloc.stride;
  a.bc = bc;
  a.bad = p->bad;
  a.x = x;
  if (p->nel > 0) {
    switch (p->kind) {
      case KIND_POISSON: hipLaunchKernelGGL(k_op_elem<KIND_POISSON>, dim3(nblk(p->nel)), dim3(256), 0, p->stream, s, a, 0.0, 0.0); break;
      case KIND_PLANE:   hipLaunchKernelGGL(k_op_elem<KIND_PLANE>, dim3(nblk(p->nel)), dim3(256), 0, p->stream, s, a, mloc.E, mloc.nu); break;
      case KIND_ELASTIC: hipLaunchKernelGGL(k_op_elem<KIND_ELASTIC>, dim3(nblk(p->nel)), dim3(256), 0, p->stream, s, a, mloc.E, mloc.nu); break;
    }
    HIPCHK(hipGetLastError());
  }
  auto gather = [&](double *out, int64_t s0, int64_t s1, int iface) {
    if (s1 <= s0) return;
    RArgs g = a;
    g.out = out;
    g.s0 = s0;
    g.s1 = s1;
    g.iface = iface;
    auto capped = [](int64_t n, int64_t cap) { return (unsigned)(n < cap ? n : cap); };
    const dim3 grid = (s.nsd == 3) ? dim3(capped((s.dim[0] * s.dim[1] + 255) / 256, 1024), 1, capped(s1 - s0, 64))
                                   : dim3(capped((s.dim[0] * (s1 - s0) + 255) / 256, 16384), 1, 1);
    switch (p->kind) {
      case KIND_POISSON: hipLaunchKernelGGL(k_rhs_gather<KIND_POISSON>, grid, dim3(256), 0, p->stream, s, g); break;
      case KIND_PLANE:   hipLaunchKernelGGL(k_rhs_gather<KIND_PLANE>, grid, dim3(256), 0, p->stream, s, g); break;
      case KIND_ELASTIC: hipLaunchKernelGGL(k_rhs_gather<KIND_ELASTIC>, grid, dim3(256), 0, p->stream, s, g); break;
    }
  };
  gather(y, s.ks, s.ke, 0);
  if (s.ks > 0 && p->send) gather(p->send + p->send_nnz, s.ks - 1, s.ks, 1);  // partial rows owed to the rank below
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(p->stream));
  return FEMASM_OK;
}

int femasm_fill_properties(femasm_plan *p, double E, double nu, int32_t stride, double *gp_out) {
  if (!p || !gp_out) return fail(FEMASM_ERR_ARG, "null argument");
  if (p->kind == KIND_POISSON) return fail(FEMASM_ERR_ARG, "the Poisson driver has no property array");
  if (p->kind == KIND_ENRICHED)
    return fail(FEMASM_ERR_ARG, "the enrichment data of an enriched record is the caller's (no driver fills it, SURVEY section 0)");
  const int need = (p->kind == KIND_PLANE) ? 24 : 400;
  if (stride < need || stride > 416) return fail(FEMASM_ERR_ARG, "stride %d not in [%d, 416]", stride, need);
  HIPCHK(hipSetDevice(p->device));
  if (p->nel <= 0) return FEMASM_OK;
  const size_t bytes = (size_t)p->nel * stride * sizeof(double);
  double *d_out = gp_out;
  const bool host = !is_device_ptr(gp_out);
  if (host) HIPCHK(hipMalloc(&d_out, bytes));
  if (p->kind == KIND_ELASTIC)
    hipLaunchKernelGGL(k_fill_props_s3, dim3(nblk(p->nel, 8)), dim3(256), 0, p->stream, p->s, p->coords, E, nu, stride,
                       p->nel, d_out);
  else
    hipLaunchKernelGGL(k_fill_props_e2, dim3(nblk(p->nel)), dim3(256), 0, p->stream, p->s, p->coords, E, nu, stride,
                       p->nel, d_out);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess && host) e = hipMemcpyAsync(gp_out, d_out, bytes, hipMemcpyDeviceToHost, p->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(p->stream);
  if (host) (void)hipFree(d_out);
  if (e != hipSuccess) return fail(FEMASM_ERR_HIP, "femasm_fill_properties: %s", hipGetErrorString(e));
  return FEMASM_OK;
}

// plan-owned scratch, grown on demand (femasm_error_integrals, femasm_coord_moments): no allocation per call
static int grow(double **buf, int64_t *have, int64_t need) {
  if (need <= *have) return FEMASM_OK;
  if (*buf) (void)hipFree(*buf);
  *buf = nullptr;
  *have = 0;
  HIPCHK(hipMalloc(buf, (size_t)need * sizeof(double)));
  *have = need;
  return FEMASM_OK;
}

int femasm_error_integrals(femasm_plan *p, const double *u, double sums[2]) {
  if (!p || !u || !sums) return fail(FEMASM_ERR_ARG, "null argument");
  if (p->kind == KIND_PLANE) return fail(FEMASM_ERR_ARG, "the reference has no error norms for the 2-D driver");
  if (p->kind == KIND_ENRICHED) return fail(FEMASM_ERR_ARG, "femasm_error_integrals: not offered for the enriched kind");
  HIPCHK(hipSetDevice(p->device));
  sums[0] = sums[1] = 0.0;
  if (p->nel <= 0) return FEMASM_OK;
  const Slab &s = p->s;
  const int64_t nghost = (int64_t)s.dim[0] * s.dim[1] * (p->cke - s.cks) * s.dof;  // u has the layers of coords
  const double *d_u = u;
  if (!is_device_ptr(u)) {
    if (int rc = grow(&p->stage_u, &p->stage_u_count, nghost)) return rc;
    HIPCHK(hipMemcpyAsync(p->stage_u, u, nghost * sizeof(double), hipMemcpyHostToDevice, p->stream));
    d_u = p->stage_u;
  }
  const int64_t nb = (p->nel + 255) / 256;
  if (int rc = grow(&p->partial, &p->partial_count, 2 * nb + 2)) return rc;
  double *partial = p->partial;
  if (p->kind == KIND_POISSON)
    hipLaunchKernelGGL(k_error_integrals<KIND_POISSON>, dim3((unsigned)nb), dim3(256), 0, p->stream, s, p->coords, d_u,
                       p->nel, partial);
  else
    hipLaunchKernelGGL(k_error_integrals<KIND_ELASTIC>, dim3((unsigned)nb), dim3(256), 0, p->stream, s, p->coords, d_u,
                       p->nel, partial);
  hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(256), 0, p->stream, partial, nb, partial + 2 * nb);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(sums, partial + 2 * nb, 2 * sizeof(double), hipMemcpyDeviceToHost, p->stream));
  HIPCHK(hipStreamSynchronize(p->stream));
  return FEMASM_OK;
}

// The entry point that allocates nothing: femasm_error_integrals for callers that cannot call hipMalloc (HIP graph
// capture, tight solver loops).  Same two kernels; the partial sums go to this device's copy of the static `scratch`
// array (address taken once per device in femasm_plan_create), the result to a DEVICE pointer, and nothing is copied
// back or waited for: no hipMalloc, hipFree, host copy or synchronisation happens here.
// Nothing in the scratch has to start at zero: k_error_integrals writes both sums of every workgroup 0 .. nb-1 before
// k_sum_partials, launched behind it on the same stream, reads exactly those nb pairs.
int femasm_error_integrals_static(femasm_plan *p, const double *u, double *sums) {
  if (!p || !u || !sums) return fail(FEMASM_ERR_ARG, "null argument");
  if (p->kind == KIND_PLANE) return fail(FEMASM_ERR_ARG, "the reference has no error norms for the 2-D driver");
  if (p->kind == KIND_ENRICHED) return fail(FEMASM_ERR_ARG, "femasm_error_integrals_static: not offered for the enriched kind");
  HIPCHK(hipSetDevice(p->device));
  if (!is_device_ptr(u) || !is_device_ptr(sums))
    return fail(FEMASM_ERR_ARG, "femasm_error_integrals_static takes DEVICE pointers (u and sums): staging would allocate");
  const int64_t nb = (p->nel + 255) / 256;
  if (nb > STATIC_NB_MAX)
    return fail(FEMASM_ERR_ARG, "femasm_error_integrals_static: %lld elements exceed the static scratch (at most %lld per slab); "
                "use femasm_error_integrals", (long long)p->nel, (long long)(256 * STATIC_NB_MAX));
  if (p->nel <= 0) {
    HIPCHK(hipMemsetAsync(sums, 0, 2 * sizeof(double), p->stream));
    return FEMASM_OK;
  }
  if (!p->static_scratch) {  // no slice was free when the plan was created: plans may have been destroyed since
    double *base = nullptr;
    if (int rcs = static_scratch_for(p->device, &base)) return rcs;
    p->static_slot = scratch_reserve(p->device, nb, &p->static_nslots);
    if (p->static_slot >= 0) p->static_scratch = base + 2 * SCR_SLOT_NB * p->static_slot;
  }
  double *partial = p->static_scratch;  // (a slice of) the looked-up device address, never the array's name
  if (!partial)
    return fail(FEMASM_ERR_ARG, "femasm_error_integrals_static: no slice of the static scratch of device %d was left for this plan "
                "(%lld workgroups; the plans alive on this GPU hold the rest): use femasm_error_integrals", p->device, (long long)nb);
  const Slab &s = p->s;
  if (p->kind == KIND_POISSON)
    hipLaunchKernelGGL(k_error_integrals<KIND_POISSON>, dim3((unsigned)nb), dim3(256), 0, p->stream, s, p->coords, u,
                       p->nel, partial);
  else
    hipLaunchKernelGGL(k_error_integrals<KIND_ELASTIC>, dim3((unsigned)nb), dim3(256), 0, p->stream, s, p->coords, u,
                       p->nel, partial);
  hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(256), 0, p->stream, partial, nb, sums);
  HIPCHK(hipGetLastError());
  return FEMASM_OK;
}

int femasm_interface_buffers(femasm_plan *p, double **send_buf, int64_t *send_count, double **recv_buf,
                             int64_t *recv_count) {
  if (!p) return fail(FEMASM_ERR_ARG, "null plan");
  if (send_buf) *send_buf = p->send;
  if (send_count) *send_count = p->send_count;
  if (recv_buf) *recv_buf = p->recv;
  if (recv_count) *recv_count = p->recv_count;
  return FEMASM_OK;
}

int femasm_interface_set_buffers(femasm_plan *p, double *send_buf, double *recv_buf) {
  if (!p) return fail(FEMASM_ERR_ARG, "null plan");
  HIPCHK(hipSetDevice(p->device));
  if ((send_buf && !is_device_ptr(send_buf)) || (recv_buf && !is_device_ptr(recv_buf)))
    return fail(FEMASM_ERR_ARG, "interface buffers must be device memory");
  HIPCHK(hipStreamSynchronize(p->stream));
  if (send_buf && p->send_count > 0) {
    if (p->own_send && p->send) (void)hipFree(p->send);
    p->send = send_buf;
    p->own_send = false;
  }
  if (recv_buf && p->recv_count > 0) {
    if (p->own_recv && p->recv) (void)hipFree(p->recv);
    p->recv = recv_buf;
    p->own_recv = false;
  }
  return FEMASM_OK;
}

int femasm_interface_exchange(femasm_plan *p, void *comm, int rank_below, int rank_above) {
  if (!p) return fail(FEMASM_ERR_ARG, "null plan");
  if (!comm) return fail(FEMASM_ERR_ARG, "null communicator (an ncclComm_t is expected)");
  const bool do_send = p->send_count > 0 && rank_below >= 0, do_recv = p->recv_count > 0 && rank_above >= 0;
  if (!do_send && !do_recv) return FEMASM_OK;
  const RcclApi &api = rccl_api();
  if (!api.ok) return fail(FEMASM_ERR_HIP, "RCCL (librccl.so: ncclSend / ncclRecv / ncclGroupStart / ncclGroupEnd) was not found");
  HIPCHK(hipSetDevice(p->device));
  // The pair goes to the plan's communication stream, behind an event that marks "the send buffer is final" on the
  // plan's stream (femasm_assemble_interface has just queued the kernels that fill it): the bulk of the assembly, queued
  // on the plan's stream after this call, runs while the transfer waits for the neighbours and moves its 21 MB.
  // femasm_interface_apply waits for the second event.
  if (!p->comm_stream) {
    HIPCHK(hipStreamCreateWithFlags(&p->comm_stream, hipStreamNonBlocking));
    HIPCHK(hipEventCreateWithFlags(&p->ev_send, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&p->ev_xchg, hipEventDisableTiming));
  }
  HIPCHK(hipEventRecord(p->ev_send, p->stream));
  HIPCHK(hipStreamWaitEvent(p->comm_stream, p->ev_send, 0));
  ncclComm_t c = (ncclComm_t)comm;
  ncclResult_t r = api.group_start();
  // one message per interface (S3:710-711 + 794-795: the stash of MatAssemblyEnd and DMLocalToGlobal's ghost sums):
  // the packed K rows and the b slice of node layer ks-1 to the rank below, the same of layer ke-1 from the rank above
  if (r == ncclSuccess && do_send) r = api.send(p->send, (size_t)p->send_count, ncclDouble, rank_below, c, p->comm_stream);
  if (r == ncclSuccess && do_recv) r = api.recv(p->recv, (size_t)p->recv_count, ncclDouble, rank_above, c, p->comm_stream);
  const ncclResult_t r2 = api.group_end();
  if (r == ncclSuccess) r = r2;
  if (r != ncclSuccess) return fail(FEMASM_ERR_HIP, "femasm_interface_exchange: RCCL: %s", api.err ? api.err(r) : "error");
  HIPCHK(hipEventRecord(p->ev_xchg, p->comm_stream));
  p->xchg_pending = true;
  return FEMASM_OK;
}

int femasm_interface_apply(femasm_plan *p, const femasm_bc *bcs, double *csr_values, double *b) {
  if (!p) return fail(FEMASM_ERR_ARG, "null plan");
  if (!p->recv) return FEMASM_OK;  // top slab: nothing is owed to us
  HIPCHK(hipSetDevice(p->device));
  if ((csr_values && !is_device_ptr(csr_values)) || (b && !is_device_ptr(b)))
    return fail(FEMASM_ERR_ARG, "femasm_interface_apply needs device-resident csr_values / b");
  const int bc = bcs ? bcs->kind : FEMASM_BC_NONE;
  const int64_t plane = (p->s.nsd == 3) ? p->s.dim[0] * p->s.dim[1] : p->s.dim[0];
  if (p->xchg_pending) {  // femasm_interface_exchange ran on the communication stream: its receive has to be complete
    HIPCHK(hipStreamWaitEvent(p->stream, p->ev_xchg, 0));
    p->xchg_pending = false;
  }
  hipLaunchKernelGGL(k_iface_apply, dim3(nblk(plane, 4)), dim3(256), 0, p->stream, p->s, bc, csr_values, b,
                     p->recv, p->recv + p->recv_nnz);
  HIPCHK(hipGetLastError());
  if (!p->async) HIPCHK(hipStreamSynchronize(p->stream));
  return FEMASM_OK;
}

int femasm_eliminate_columns(femasm_plan *p, const femasm_bc *bcs, double *csr_values, double *b) {
  if (!p || !csr_values) return fail(FEMASM_ERR_ARG, "null argument");
  const int bc = bcs ? bcs->kind : FEMASM_BC_NONE;
  if (bc == FEMASM_BC_NONE) return FEMASM_OK;
  if (p->kind == KIND_ENRICHED) return fail(FEMASM_ERR_ARG, "femasm_eliminate_columns: not offered for the enriched kind");
  if ((bc == FEMASM_BC_POISSON_INLOOP && p->kind != KIND_POISSON) || (bc == FEMASM_BC_E2_COMPRESSION && p->kind != KIND_PLANE) ||
      (bc == FEMASM_BC_S3_ALLFACES_EXACT && p->kind != KIND_ELASTIC) || bc < 0 || bc > 3)
    return fail(FEMASM_ERR_ARG, "bc kind %d does not belong to problem kind %d", bc, p->kind);
  HIPCHK(hipSetDevice(p->device));
  if (!is_device_ptr(csr_values) || (b && !is_device_ptr(b)))
    return fail(FEMASM_ERR_ARG, "femasm_eliminate_columns works on device-resident csr_values / b");
  const Slab &s = p->s;
  const int64_t SL = (s.nsd == 3) ? s.dim[2] : s.dim[1];
  if (bc == FEMASM_BC_S3_ALLFACES_EXACT && p->cke < (s.ke + 1 < SL ? s.ke + 1 : SL))
    return fail(FEMASM_ERR_ARG, "the coordinates must include node layer ke (the Dirichlet values of the columns above the slab)");
  hipLaunchKernelGGL(k_eliminate_columns, dim3(nblk(p->nrows, 4)), dim3(256), 0, p->stream, s, bc, p->coords, p->nrows,
                     p->rowptr, p->colidx, csr_values, b);
  HIPCHK(hipGetLastError());
  if (!p->async) HIPCHK(hipStreamSynchronize(p->stream));
  return FEMASM_OK;
}

int femasm_compress_bc_rows(femasm_plan *p, const femasm_bc *bcs, const double *csr_values, int64_t *rowptr_out,
                            int32_t *colidx_out, double *values_out, int64_t *nnz_out) {
  if (!p || !csr_values || !rowptr_out || !colidx_out || !values_out || !nnz_out)
    return fail(FEMASM_ERR_ARG, "null argument");
  HIPCHK(hipSetDevice(p->device));
  const int bc = bcs ? bcs->kind : FEMASM_BC_NONE;
  // MatZeroRows is what the elastic drivers call (E2:794-799, S3:1513-1518); the Poisson driver never does: its
  // boundary rows are assembled (diagonal = number of elements sharing the node, P3:286-340) and stay in the pattern
  if (p->kind == KIND_ENRICHED)
    return fail(FEMASM_ERR_ARG, "femasm_compress_bc_rows: not offered for the enriched kind (blocked dofs depend on the records)");
  if (bc == FEMASM_BC_POISSON_INLOOP)
    return fail(FEMASM_ERR_ARG, "femasm_compress_bc_rows: the Poisson driver has no MatZeroRows step (its Dirichlet "
                "rows are assembled in the element loop, diagonal = element valence)");
  const bool in_dev = is_device_ptr(csr_values);
  const bool out_dev = is_device_ptr(rowptr_out) && is_device_ptr(colidx_out) && is_device_ptr(values_out);
  const bool out_host = !is_device_ptr(rowptr_out) && !is_device_ptr(colidx_out) && !is_device_ptr(values_out);
  if (!out_dev && !out_host) return fail(FEMASM_ERR_ARG, "outputs must be all host or all device");

  const int64_t nrows = p->nrows, nnz = p->nnz;
  int rc = FEMASM_OK;
  double *d_in = nullptr, *d_vals = nullptr;
  int64_t *d_len = nullptr, *d_rp = nullptr;
  int32_t *d_ci = nullptr;
  void *d_tmp = nullptr;
  size_t tmp_bytes = 0;
  hipError_t e = hipSuccess;
#define CCHK(call) do { if (e == hipSuccess) e = (call); } while (0)
  if (!in_dev) {
    CCHK(hipMalloc(&d_in, nnz * sizeof(double)));
    CCHK(hipMemcpy(d_in, csr_values, nnz * sizeof(double), hipMemcpyHostToDevice));
  }
  const double *vin = in_dev ? csr_values : d_in;
  CCHK(hipMalloc(&d_len, (nrows + 1) * sizeof(int64_t)));
  if (out_dev) { d_rp = rowptr_out; d_ci = colidx_out; d_vals = values_out; }
  else {
    CCHK(hipMalloc(&d_rp, (nrows + 1) * sizeof(int64_t)));
    CCHK(hipMalloc(&d_ci, nnz * sizeof(int32_t)));
    CCHK(hipMalloc(&d_vals, nnz * sizeof(double)));
  }
  if (e == hipSuccess) {
    hipLaunchKernelGGL(k_bc_rowlen, dim3(nblk(nrows + 1)), dim3(256), 0, p->stream, p->s, bc, nrows, p->rowptr, d_len);
    CCHK(hipGetLastError());
    CCHK(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, d_len, d_rp, (int)(nrows + 1), p->stream));
    CCHK(hipMalloc(&d_tmp, tmp_bytes ? tmp_bytes : 1));
    CCHK(hipcub::DeviceScan::ExclusiveSum(d_tmp, tmp_bytes, d_len, d_rp, (int)(nrows + 1), p->stream));
  }
  if (e == hipSuccess) {
    hipLaunchKernelGGL(k_bc_compress, dim3(nblk(nrows)), dim3(256), 0, p->stream, p->s, bc, nrows, p->rowptr,
                       p->colidx, vin, d_rp, d_ci, d_vals);
    CCHK(hipGetLastError());
  }
  int64_t newnnz = 0;
  CCHK(hipMemcpyAsync(&newnnz, d_rp + nrows, sizeof(int64_t), hipMemcpyDeviceToHost, p->stream));
  CCHK(hipStreamSynchronize(p->stream));
  if (e == hipSuccess && out_host) {
    CCHK(hipMemcpy(rowptr_out, d_rp, (nrows + 1) * sizeof(int64_t), hipMemcpyDeviceToHost));
    CCHK(hipMemcpy(colidx_out, d_ci, newnnz * sizeof(int32_t), hipMemcpyDeviceToHost));
    CCHK(hipMemcpy(values_out, d_vals, newnnz * sizeof(double), hipMemcpyDeviceToHost));
  }
#undef CCHK
  if (d_in) (void)hipFree(d_in);
  if (d_len) (void)hipFree(d_len);
  if (d_tmp) (void)hipFree(d_tmp);
  if (out_host) {
    if (d_rp) (void)hipFree(d_rp);
    if (d_ci) (void)hipFree(d_ci);
    if (d_vals) (void)hipFree(d_vals);
  }
  if (e != hipSuccess) rc = fail(FEMASM_ERR_HIP, "femasm_compress_bc_rows: %s", hipGetErrorString(e));
  *nnz_out = newnnz;
  return rc;
}

int femasm_coo_indices(femasm_plan *p, int32_t index_bytes, void *coo_i, void *coo_j) {
  if (!p || !coo_i || !coo_j) return fail(FEMASM_ERR_ARG, "null argument");
  if (index_bytes != 4 && index_bytes != 8) return fail(FEMASM_ERR_ARG, "index_bytes must be 4 or 8 (PetscInt)");
  HIPCHK(hipSetDevice(p->device));
  const bool dev = is_device_ptr(coo_i);
  if (dev != is_device_ptr(coo_j)) return fail(FEMASM_ERR_ARG, "coo_i and coo_j must be both host or both device");
  const Slab &s = p->s;
  const int64_t plane = (s.nsd == 3) ? s.dim[0] * s.dim[1] : s.dim[0];
  const int64_t row0 = plane * s.ks * s.dof;
  const int64_t ntot = plane * (s.nsd == 3 ? s.dim[2] : s.dim[1]) * s.dof;
  if (index_bytes == 4 && ntot > INT32_MAX) return fail(FEMASM_ERR_ARG, "global row ids need 64-bit indices");
  const size_t bytes = (size_t)p->nnz * index_bytes;
  void *di = coo_i, *dj = coo_j;
  if (!dev) {
    HIPCHK(hipMalloc(&di, bytes));
    if (hipMalloc(&dj, bytes) != hipSuccess) { (void)hipFree(di); return fail(FEMASM_ERR_NOMEM, "coo staging"); }
  }
  if (index_bytes == 4)
    hipLaunchKernelGGL(k_coo_indices<int32_t>, dim3(nblk(p->nrows, 4)), dim3(256), 0, p->stream, p->nrows, row0,
                       p->rowptr, p->colidx, (int32_t *)di, (int32_t *)dj);
  else
    hipLaunchKernelGGL(k_coo_indices<int64_t>, dim3(nblk(p->nrows, 4)), dim3(256), 0, p->stream, p->nrows, row0,
                       p->rowptr, p->colidx, (int64_t *)di, (int64_t *)dj);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess && !dev) e = hipMemcpyAsync(coo_i, di, bytes, hipMemcpyDeviceToHost, p->stream);
  if (e == hipSuccess && !dev) e = hipMemcpyAsync(coo_j, dj, bytes, hipMemcpyDeviceToHost, p->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(p->stream);
  if (!dev) { (void)hipFree(di); (void)hipFree(dj); }
  if (e != hipSuccess) return fail(FEMASM_ERR_HIP, "femasm_coo_indices: %s", hipGetErrorString(e));
  return FEMASM_OK;
}

int femasm_coord_moments(femasm_plan *p, double mom[10]) {
  if (!p || !mom) return fail(FEMASM_ERR_ARG, "null argument");
  HIPCHK(hipSetDevice(p->device));
  const Slab &s = p->s;
  const int64_t nown = p->nrows / s.dof;
  for (int q = 0; q < 10; q++) mom[q] = 0.0;
  if (nown <= 0) return FEMASM_OK;
  int64_t nb = (nown + 255) / 256;
  if (nb > 1024) nb = 1024;
  if (int rc = grow(&p->partial, &p->partial_count, 10 * nb + 10)) return rc;
  double *partial = p->partial;
  hipLaunchKernelGGL(k_coord_moments, dim3((unsigned)nb), dim3(256), 0, p->stream, s, p->coords, nown, partial);
  hipLaunchKernelGGL(k_sum_rows10, dim3(1), dim3(64), 0, p->stream, partial, nb, partial + 10 * nb);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(mom, partial + 10 * nb, 10 * sizeof(double), hipMemcpyDeviceToHost, p->stream));
  HIPCHK(hipStreamSynchronize(p->stream));
  return FEMASM_OK;
}

int femasm_rigid_body_modes(femasm_plan *p, const double mom[10], double *modes) {
  if (!p || !mom || !modes) return fail(FEMASM_ERR_ARG, "null argument");
  if (p->kind == KIND_POISSON || p->kind == KIND_ENRICHED)
    return fail(FEMASM_ERR_ARG, "rigid-body modes belong to the elasticity kinds with 2 or 3 dofs per node");
  if (!(mom[0] >= 1.0)) return fail(FEMASM_ERR_ARG, "mom[0] (global node count) must be >= 1");
  HIPCHK(hipSetDevice(p->device));
  const Slab &s = p->s;
  const int nsd = s.nsd, nm = (nsd == 3) ? 6 : 3;
  const double sN = 1.0 / sqrt(mom[0]);
  const double Sx = mom[1], Sy = mom[2], Sz = mom[3], Sxx = mom[4], Syy = mom[5], Szz = mom[6], Sxy = mom[7],
               Syz = mom[8], Szx = mom[9];
  // Gram matrix of the raw vectors
  double G[6][6] = {};
  for (int k = 0; k < nsd; k++) G[k][k] = mom[0] * sN * sN;
  if (nsd == 3) {
    G[0][3] = sN * Sy;  G[1][3] = -sN * Sx;
    G[1][4] = -sN * Sz; G[2][4] = sN * Sy;
    G[0][5] = sN * Sz;  G[2][5] = -sN * Sx;
    G[3][3] = Syy + Sxx; G[4][4] = Szz + Syy; G[5][5] = Szz + Sxx;
    G[3][4] = Szx; G[3][5] = Syz; G[4][5] = -Sxy;
  } else {
    G[0][2] = -sN * Sy; G[1][2] = sN * Sx; G[2][2] = Syy + Sxx;
  }
  for (int i = 0; i < 6; i++)
    for (int j = 0; j < i; j++) G[i][j] = G[j][i];
  // classical Gram-Schmidt of vector i against the finished 0..i-1, then normalise -- in coefficient space
  RigidArgs a = {};
  a.sN = sN;
  a.nmodes = nm;
  for (int k = 0; k < nsd; k++) a.C[k][k] = 1.0;
  for (int i = nsd; i < nm; i++) {
    double u[6] = {};
    u[i] = 1.0;
    for (int j = 0; j < i; j++) {
      double dot = 0.0;
      for (int k = 0; k < nm; k++) dot += a.C[j][k] * G[k][i];
      for (int k = 0; k < nm; k++) u[k] -= dot * a.C[j][k];
    }
    double n2 = 0.0;
    for (int k = 0; k < nm; k++)
      for (int l = 0; l < nm; l++) n2 += u[k] * G[k][l] * u[l];
    if (!(n2 > 0.0)) return fail(FEMASM_ERR_GEOMETRY, "rigid-body mode %d is degenerate (all nodes on a line?)", i);
    const double inv = 1.0 / sqrt(n2);
    for (int k = 0; k < nm; k++) a.C[i][k] = u[k] * inv;
  }
  const int64_t nown = p->nrows / s.dof;
  if (nown <= 0) return FEMASM_OK;
  const bool dev = is_device_ptr(modes);
  const size_t bytes = (size_t)nm * p->nrows * sizeof(double);
  double *d = modes;
  if (!dev) HIPCHK(hipMalloc(&d, bytes));
  hipLaunchKernelGGL(k_rigid_modes, dim3(nblk(nown)), dim3(256), 0, p->stream, s, p->coords, nown, a, d);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess && !dev) e = hipMemcpyAsync(modes, d, bytes, hipMemcpyDeviceToHost, p->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(p->stream);
  if (!dev) (void)hipFree(d);
  if (e != hipSuccess) return fail(FEMASM_ERR_HIP, "femasm_rigid_body_modes: %s", hipGetErrorString(e));
  return FEMASM_OK;
}

int femasm_zero_rows(femasm_plan *p, int64_t n, const int64_t *rows, const double *vals, double *csr_values,
                     double *b) {
  if (!p || (n > 0 && !rows)) return fail(FEMASM_ERR_ARG, "null argument");
  if (n <= 0 || (!csr_values && !b)) return FEMASM_OK;
  HIPCHK(hipSetDevice(p->device));
  if ((csr_values && !is_device_ptr(csr_values)) || (b && !is_device_ptr(b)))
    return fail(FEMASM_ERR_ARG, "femasm_zero_rows works on device-resident csr_values / b");
  const Slab &s = p->s;
  const int64_t plane = (s.nsd == 3) ? s.dim[0] * s.dim[1] : s.dim[0];
  const int64_t row0 = plane * s.ks * s.dof;
  // the list (and the values) may live on the host: stage them
  int64_t *d_rows = const_cast<int64_t *>(rows);
  double *d_vals = const_cast<double *>(vals);
  bool own_rows = false, own_vals = false;
  hipError_t e = hipSuccess;
  if (!is_device_ptr(rows)) {
    HIPCHK(hipMalloc(&d_rows, n * sizeof(int64_t)));  // (nothing else is held yet)
    own_rows = true;
    e = hipMemcpyAsync(d_rows, rows, n * sizeof(int64_t), hipMemcpyHostToDevice, p->stream);
  }
  if (e == hipSuccess && vals && !is_device_ptr(vals)) {
    e = hipMalloc(&d_vals, n * sizeof(double));
    if (e == hipSuccess) {
      own_vals = true;
      e = hipMemcpyAsync(d_vals, vals, n * sizeof(double), hipMemcpyHostToDevice, p->stream);
    }
  }
  if (e == hipSuccess) {
    hipLaunchKernelGGL(k_zero_rows, dim3(nblk(n, 4)), dim3(256), 0, p->stream, n, d_rows, d_vals, row0, p->nrows,
                       p->rowptr, p->colidx, csr_values, b);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipStreamSynchronize(p->stream);
  if (own_rows) (void)hipFree(d_rows);
  if (own_vals) (void)hipFree(d_vals);
  if (e != hipSuccess) return fail(FEMASM_ERR_HIP, "femasm_zero_rows: %s", hipGetErrorString(e));
  return FEMASM_OK;
}

}  // extern "C"
