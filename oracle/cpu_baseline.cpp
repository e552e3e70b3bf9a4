// ORACLE (test infrastructure, NOT product code): CPU baseline driver of bench.py's `cpu_baseline` leg.
//
// Times the CPU restatement of the reference path (oracle/*.hpp: sparse assembly in the reference's row order + the
// Clarabel-style IPM at the reference's tolerances) on the SAME workload as the GPU bench -- Config B instances, 10
// cold-start solves each (MPC::CreateInitialRun, untimed) then `steps` open-loop RTI iterations
// (/root/reference/test/gait_opt_playground.cpp:113-126) -- as BASELINE.md section 3.4 planned it: compiled
// -O3 -march=native -ffp-contract=off ON THE BOX IT RUNS ON, (i) one thread, (ii) OpenMP over instances on all host cores.
//
//   cpu_baseline <input.bin> <n_single> <n_per_thread> <steps>
// input.bin (float64, written by bench.py): [num_nodes, dt, friction, force_bound, swing_height, foot_offset, box_x, box_y,
//   force_cost, mass, Ir[9], hip_xy[8], Q_diag[12], des_state[13], n_inst, then n_inst x (state[13], ee[12])]
// Prints one JSON object.
#include <omp.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <thread>
#include <vector>
#include "srbm_gait.hpp"

using namespace orc;
using clk = std::chrono::steady_clock;

struct Input {
    MPCInfo info;
    SRBModel model;
    double Q[144] = {0};
    Vec13 des;
    std::vector<Vec13> states;
    std::vector<std::vector<Vec3>> ees;
};

static bool read_input(const char* path, Input& in) {
    std::ifstream f(path, std::ios::binary | std::ios::ate);
    if (!f) return false;
    const size_t n = (size_t)f.tellg() / sizeof(double);
    std::vector<double> v(n);
    f.seekg(0);
    f.read(reinterpret_cast<char*>(v.data()), n * sizeof(double));
    size_t o = 0;
    auto next = [&]() { return v.at(o++); };
    in.info.num_nodes = (int)next(); in.info.integrator_dt = next(); in.info.friction_coef = next(); in.info.force_bound = next();
    in.info.swing_height = next(); in.info.foot_offset = next(); in.info.ee_box_size[0] = next(); in.info.ee_box_size[1] = next();
    in.info.force_cost = next();
    in.model.mass = next();
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) in.model.Ir.m[i][j] = next();
    in.model.Ir_inv = in.model.Ir.inverse();
    for (int e = 0; e < 4; e++) { in.model.hip_xy[e][0] = next(); in.model.hip_xy[e][1] = next(); }
    for (int i = 0; i < 12; i++) in.Q[13 * i] = next();
    for (int i = 0; i < 13; i++) in.des[i] = next();
    const int ninst = (int)next();
    for (int b = 0; b < ninst; b++) {
        Vec13 s;
        for (int i = 0; i < 13; i++) s[i] = next();
        std::vector<Vec3> ee(4);
        for (int e = 0; e < 4; e++) ee[e] = {next(), next(), next()};
        in.states.push_back(s); in.ees.push_back(ee);
    }
    return true;
}

// one instance: set-up as controllers/mpc_controller.cpp:57-67, cold start, then `steps` timed RTI iterations.
// returns seconds of the timed part; *iters accumulates the IPM iterations of the timed solves
static double run_instance(const Input& in, int b, int steps, long* iters, int* bad) {
    MPCSingleRigidBody mpc(in.info, in.model);
    const Vec12 des = SRBModel::ManifoldToTangent(in.des);
    mpc.AddQuadraticTrackingCost(des.data(), in.Q);
    mpc.SetQuadraticFinalCost(in.Q);
    double w[12];
    for (int i = 0; i < 12; i++) w[i] = -1 * in.Q[13 * i] * des[i];
    mpc.SetLinearFinalCost(w);
    mpc.SetStateTrajectoryWarmStart(std::vector<Vec13>(in.info.num_nodes + 1, in.states[b]));
    mpc.CreateInitialRun(in.states[b], in.ees[b]);
    Vec13 state = in.states[b];
    const auto t0 = clk::now();
    for (int i = 0; i < steps; i++) {
        const double t = i * in.info.integrator_dt;
        std::vector<Vec3> ee(4);
        for (int e = 0; e < 4; e++)
            for (int c = 0; c < 3; c++) ee[e][c] = mpc.GetTrajectory().EE(e).ValueAt(orc::Position, c, t);
        mpc.GetRealTimeUpdate(state, t, ee);
        state = mpc.GetTrajectory().GetState(1);
        *iters += mpc.Stats().qp_iters;
        if (mpc.GetSolveQuality() != Solved && mpc.GetSolveQuality() != SolvedInacc) (*bad)++;
    }
    return std::chrono::duration<double>(clk::now() - t0).count();
}

int main(int argc, char** argv) {
    if (argc < 5) { std::fprintf(stderr, "usage: cpu_baseline <input.bin> <n_single> <n_per_thread> <steps>\n"); return 2; }
    Input in;
    if (!read_input(argv[1], in)) { std::fprintf(stderr, "cannot read %s\n", argv[1]); return 2; }
    const int n_single = std::atoi(argv[2]), per_thread = std::atoi(argv[3]), steps = std::atoi(argv[4]);
    const int ninst = (int)in.states.size();
    // (i) one thread
    double el1 = 0; long it1 = 0; int bad1 = 0;
    for (int b = 0; b < n_single && b < ninst; b++) el1 += run_instance(in, b, steps, &it1, &bad1);
    const int done1 = std::min(n_single, ninst);
    // (ii) OpenMP over instances on all host cores: every thread runs `per_thread` instances start to finish; the wall clock of the
    // timed parts is taken as the slowest thread's (cold starts are set-up on both sides of the comparison)
    const int nthreads = omp_get_max_threads();
    std::vector<double> tt(nthreads, 0.0);
    long itN = 0; int badN = 0;
    const auto w0 = clk::now();
    #pragma omp parallel num_threads(nthreads) reduction(+ : itN, badN)
    {
        const int k = omp_get_thread_num();
        for (int j = 0; j < per_thread; j++) tt[k] += run_instance(in, (n_single + k * per_thread + j) % ninst, steps, &itN, &badN);
    }
    const double wall = std::chrono::duration<double>(clk::now() - w0).count();
    double slow = 0;
    for (double v : tt) slow = std::max(slow, v);
    std::printf("{\"single\": {\"it_per_s\": %.6g, \"instances\": %d, \"steps\": %d, \"seconds\": %.6g, \"mean_ipm_iterations\": %.4g, \"not_solved\": %d}, "
                "\"all_cores\": {\"it_per_s\": %.6g, \"threads\": %d, \"instances\": %d, \"steps\": %d, \"slowest_thread_seconds\": %.6g, "
                "\"wall_s_incl_cold_starts\": %.6g, \"mean_ipm_iterations\": %.4g, \"not_solved\": %d}, \"hardware_threads\": %u}\n",
                done1 * steps / el1, done1, steps, el1, (double)it1 / std::max(1, done1 * steps), bad1,
                (double)nthreads * per_thread * steps / slow, nthreads, nthreads * per_thread, steps, slow, wall,
                (double)itN / std::max(1, nthreads * per_thread * steps), badN, std::thread::hardware_concurrency());
    return 0;
}
