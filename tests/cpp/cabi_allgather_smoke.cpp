// The C side of the multi-GPU collective (include/srbm_rti.h: srbm_allgather_results) driven from a C++ host that links RCCL ITSELF, as the
// MPC thread of a controller process would (one process per GPU; controllers/mpc_controller.cpp:286-399 is the host that would call it): the
// host makes its own communicator with ncclCommInitRank, hands the plain ncclComm_t to the library, and reads the gathered records from the
// device buffer.  World size 1 here (a one-GPU box): the gathered array must equal srbm_pack_results bit for bit.  The second half does the
// same through the library's own communicator helpers (hosts that do not link RCCL: ctypes, bench.py).
//
//   g++ cabi_allgather_smoke.cpp -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -lsrbm_rti -lrccl -lamdhip64
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>
#include "srbm_rti.h"
#include "cfg.inc"

static void check(int rc) { if (rc != 0) { std::fprintf(stderr, "srbm: %s\n", srbm_last_error()); std::exit(1); } }
static void hipok(hipError_t e) { if (e != hipSuccess) { std::fprintf(stderr, "hip: %s\n", hipGetErrorString(e)); std::exit(1); } }
static void ncclok(ncclResult_t r) { if (r != ncclSuccess) { std::fprintf(stderr, "rccl: %s\n", ncclGetErrorString(r)); std::exit(1); } }

int main() {
    srbm_mpc_info info{};
    info.num_nodes = kNumNodes; info.integrator_dt = kDt; info.friction_coef = kMu; info.force_bound = kForceBound;
    info.swing_height = kSwing; info.foot_offset = kFootOffset; info.ee_box_size[0] = kBox[0]; info.ee_box_size[1] = kBox[1]; info.force_cost = kForceCost;
    srbm_model model{};
    model.mass = kMass;
    for (int i = 0; i < 9; i++) model.Ir[i] = kIr[i];
    for (int i = 0; i < 8; i++) model.hip_xy[i] = kHip[i];
    const int B = 3;
    srbm_batch* h = nullptr;
    check(srbm_batch_create(&h, B, &info, &model, 0));
    std::vector<double> Q(144, 0.0), des(kTargetTangent, kTargetTangent + 12), w(12, 0.0);
    for (int i = 0; i < 12; i++) { Q[i * 13] = kQdiag[i]; w[i] = -kQdiag[i] * des[i]; }
    check(srbm_add_quadratic_tracking_cost(h, des.data(), Q.data()));
    check(srbm_set_quadratic_final_cost(h, Q.data()));
    check(srbm_set_linear_final_cost(h, w.data()));
    std::vector<double> state(13 * B), ee(12 * B);
    const double ee0[12] = {0.2, 0.2, 0, 0.2, -0.2, 0, -0.2, 0.2, 0, -0.2, -0.2, 0};
    for (int b = 0; b < B; b++) {
        for (int i = 0; i < 13; i++) state[13 * b + i] = kInit[i];
        state[13 * b] += 0.01 * b;                                   // three different instances
        for (int i = 0; i < 12; i++) ee[12 * b + i] = ee0[i];
    }
    check(srbm_set_state_trajectory_warm_start(h, state.data()));
    check(srbm_create_initial_run(h, state.data(), ee.data()));
    check(srbm_rti_advance(h, 0, 2));
    check(srbm_synchronize(h));

    const int ld = srbm_result_record_doubles(kNumNodes);
    const size_t n = (size_t)B * ld;
    std::vector<double> packed(n), gathered(n), gathered2(n);
    check(srbm_pack_results(h, packed.data(), ld));

    // (1) the host's own communicator
    hipok(hipSetDevice(0));
    ncclUniqueId id;
    ncclComm_t comm = nullptr;
    ncclok(ncclGetUniqueId(&id));
    ncclok(ncclCommInitRank(&comm, 1, id, 0));
    double* out = nullptr;
    hipok(hipMalloc(reinterpret_cast<void**>(&out), n * sizeof(double)));
    hipok(hipMemset(out, 0xff, n * sizeof(double)));
    check(srbm_allgather_results(h, comm, out));
    check(srbm_synchronize(h));
    hipok(hipMemcpy(gathered.data(), out, n * sizeof(double), hipMemcpyDeviceToHost));
    ncclok(ncclCommDestroy(comm));

    // (2) a communicator made by the library's helpers
    unsigned char idb[SRBM_RCCL_UNIQUE_ID_BYTES];
    ncclComm_t comm2 = nullptr;
    check(srbm_rccl_get_unique_id(idb));
    check(srbm_rccl_comm_init_rank(h, 1, 0, idb, &comm2));
    hipok(hipMemset(out, 0xff, n * sizeof(double)));
    check(srbm_allgather_results(h, comm2, out));
    check(srbm_synchronize(h));
    hipok(hipMemcpy(gathered2.data(), out, n * sizeof(double), hipMemcpyDeviceToHost));
    check(srbm_rccl_comm_destroy(comm2));
    hipok(hipFree(out));

    std::printf("record_doubles 0 %d\n", ld);
    std::printf("own_comm_equal 0 %d\n", (int)(std::memcmp(packed.data(), gathered.data(), n * sizeof(double)) == 0));
    std::printf("helper_comm_equal 0 %d\n", (int)(std::memcmp(packed.data(), gathered2.data(), n * sizeof(double)) == 0));
    for (int b = 0; b < B; b++) std::printf("status %d %d\n", b, (int)gathered[(size_t)b * ld]);
    for (int b = 0; b < B; b++) std::printf("x0 %d %.17g\n", b, gathered[(size_t)b * ld + 8]);
    check(srbm_batch_destroy(h));
    return 0;
}
