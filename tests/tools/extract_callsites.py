"""Builds, AT TEST TIME and in the build container only, translation units that hold the reference's own caller code -- cut out of
/root/reference as text, never stored in this repository -- between a builder-written prologue (includes, the two visualiser types and the
Catch2 / timer names the callers mention, a class shell with the members the controller's methods touch) and compiles them against
include/mpc_facade/mpc.h.  What it checks is SURVEY.md section 8(b): the callers of the hot path compile UNCHANGED against the facade.

  playground unit   test/gait_opt_playground.cpp: RunGaitOpt, PrintContactSched, MPCWithFixedPosition (whole functions, by brace matching)
                    test/mpc_test.cpp: the body of SECTION("Model Partials") as the body of a function with the section's free variables as arguments
  controller unit   controllers/mpc_controller.cpp: MPCController::MPCUpdate and MPCController::GaitOpt as members of a class shell

    write_units(reference_root, out_dir) -> [paths]      (raises FileNotFoundError when the reference is not there: the test then skips)
"""
import os
import re


def _function_text(src, start_regex):
    """text of the function whose first line matches start_regex, through its closing brace"""
    m = re.search(start_regex, src, re.M)
    if not m:
        raise ValueError('not found: ' + start_regex)
    # the first '{' AFTER the parameter list: skip to the ')' that closes it
    depth, k = 0, m.start()
    while True:
        ch = src[k]
        if ch == '(':
            depth += 1
        elif ch == ')':
            depth -= 1
            if depth == 0:
                break
        k += 1
    i = src.index('{', k)
    return src[m.start():_match_brace(src, i) + 1]


def _match_brace(src, i):
    """index of the '}' that closes the '{' at i (comments, strings and character literals skipped)"""
    assert src[i] == '{'
    depth, k, n = 0, i, len(src)
    while k < n:
        two = src[k:k + 2]
        if two == '//':
            k = src.index('\n', k)
            continue
        if two == '/*':
            k = src.index('*/', k) + 2
            continue
        ch = src[k]
        if ch in '"\'':
            q = ch
            k += 1
            while src[k] != q:
                k += 2 if src[k] == '\\' else 1
        elif ch == '{':
            depth += 1
        elif ch == '}':
            depth -= 1
            if depth == 0:
                return k
        k += 1
    raise ValueError('unbalanced braces')


def _section_body(src, name):
    m = re.search(r'SECTION\("%s"\)\s*\{' % re.escape(name), src)
    if not m:
        raise ValueError('section not found: ' + name)
    i = m.end() - 1
    return src[i + 1:_match_brace(src, i)]


PLAYGROUND_PROLOGUE = r'''// GENERATED at test time by tests/tools/extract_callsites.py -- prologue and epilogue are the builder's, everything between the BEGIN / END
// markers is the reference's text, read from its checkout and not kept in this repository.
#include <cassert>
#include <cmath>
#include <iomanip>
#include <iostream>
#include <memory>
#include <string>
#include <vector>

#include "mpc_facade/mpc.h"

using vector_t = Eigen::VectorXd;
using matrix_t = Eigen::MatrixXd;

// the two types of the MuJoCo visualiser the playground's signature mentions (out of scope, SURVEY.md section 2): empty shells
namespace simulator { struct SimulationRobot { vector_t ConvertPinocchioConfigToMujoco(const vector_t& q) const { return q; } }; }
namespace simulation {
struct Visualizer {
    void UpdateState(const vector_t&) {}
    template <class A, class B, class C> void GetTrajViz(const A&, const B&, const C&) {}
    void UpdateViz(double) {}
};
}
'''

PARTIALS_PROLOGUE = r'''
// the Catch2 names SECTION("Model Partials") uses, as plain checks
namespace Catch { namespace Matchers {
struct WithinAbsMatcher { double target, margin; bool match(double v) const { return std::abs(v - target) <= margin; } };
inline WithinAbsMatcher WithinAbs(double target, double margin) { return WithinAbsMatcher{target, margin}; }
} }
static int g_failed_requirements = 0;
#define REQUIRE(cond) do { if (!(cond)) g_failed_requirements++; } while (0)
#define REQUIRE_THAT(value, matcher) do { if (!(matcher).match(value)) g_failed_requirements++; } while (0)
using namespace mpc;

void ModelPartialsSection(mpc::MPCSingleRigidBody& mpc, mpc::MPCSingleRigidBody& mpc2, const vector_t& init_state,
                          std::vector<Eigen::Vector3d>& ee_locations) {
    using Catch::Matchers::WithinAbs;
'''

CONTROLLER_PROLOGUE = r'''// GENERATED at test time by tests/tools/extract_callsites.py -- the class shell is the builder's (the members MPCUpdate / GaitOpt touch, typed as
// controllers/include/mpc_controller.h:80-131 types them), the two member functions between the markers are the reference's text.
#include <cmath>
#include <fstream>
#include <iostream>
#include <mutex>
#include <string>
#include <tuple>
#include <vector>

#include "mpc_facade/mpc.h"

namespace utils {
class Timer {              // utils/include/timer.h:13-34: the calls the controller makes, as no-ops
public:
    explicit Timer(std::string) {}
    void StartTimer() {}
    void StopTimer() {}
    void PrintElapsedTime() const {}
};
}
namespace controller {
using vector_t = Eigen::VectorXd;
using matrix_t = Eigen::MatrixXd;
class MPCController {
public:
    MPCController(const mpc::MPCInfo& info, const srbm_model& consts) : mpc_(info, consts), gait_opt_(4, 10, 10, 10, 1, 0.05), info_(info) {}
    void UpdateTrajViz() {}
    void MPCUpdate();
    bool GaitOpt(double cost_red, double time, const std::vector<Eigen::Vector3d>& ee_locations);
    void PrintContactTimes() const {}
    mpc::MPCSingleRigidBody mpc_;
    mpc::GaitOptimizer gait_opt_;
    int gait_opt_freq_ = 5;
    vector_t state_;
    double time_ = 0;
    mpc::Trajectory traj_;
    std::mutex state_time_mut_, mpc_res_mut_, traj_viz_mut_, sync_mut_, model_mut_;
    std::vector<mpc::vector_3t> ee_locations_;
    mpc::MPCInfo info_;
    int run_num = 0;
    Contact contact_;
    std::ofstream log_file_;
};
'''


def write_units(reference_root, out_dir):
    pg = os.path.join(reference_root, 'test', 'gait_opt_playground.cpp')
    mt = os.path.join(reference_root, 'test', 'mpc_test.cpp')
    mc = os.path.join(reference_root, 'controllers', 'mpc_controller.cpp')
    for p in (pg, mt, mc):
        if not os.path.exists(p):
            raise FileNotFoundError(p)
    pg_src, mt_src, mc_src = open(pg).read(), open(mt).read(), open(mc).read()
    begin, end = '// ---- BEGIN reference text: %s ----\n', '\n// ---- END reference text ----\n'
    unit1 = PLAYGROUND_PROLOGUE
    for rx in (r'^double RunGaitOpt\(', r'^void PrintContactSched\(', r'^void MPCWithFixedPosition\('):
        unit1 += begin % ('test/gait_opt_playground.cpp ' + rx) + _function_text(pg_src, rx) + end
    unit1 += PARTIALS_PROLOGUE + begin % 'test/mpc_test.cpp SECTION("Model Partials")' + _section_body(mt_src, 'Model Partials') + end + '}\n'
    unit1 += '\nint main() { return g_failed_requirements; }\n'
    unit2 = CONTROLLER_PROLOGUE
    for rx in (r'^\s*void MPCController::MPCUpdate\(\)', r'^\s*bool MPCController::GaitOpt\('):
        unit2 += begin % ('controllers/mpc_controller.cpp ' + rx) + _function_text(mc_src, rx) + end
    unit2 += '}  // namespace controller\n\nint main() { return 0; }\n'
    paths = []
    for name, text in (('reference_playground_unit.cpp', unit1), ('reference_controller_unit.cpp', unit2)):
        p = os.path.join(out_dir, name)
        with open(p, 'w') as f:
            f.write(text)
        paths.append(p)
    return paths


if __name__ == '__main__':
    import sys
    print('\n'.join(write_units(sys.argv[1] if len(sys.argv) > 1 else '/root/reference', sys.argv[2] if len(sys.argv) > 2 else '.')))
