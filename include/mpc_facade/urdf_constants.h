// The constants mpc::MPC's constructor asks pinocchio for, from the URDF it is given
// (/root/reference/mpc/models/model.cpp:14-37, mpc/models/single_rigid_body_model.cpp:19-42,258-308):
//   mass   = pinocchio::computeTotalMass                                  (sum of the link masses)
//   Ir     = composite rigid-body rotational inertia of the whole robot at the nominal configuration, about the whole-body
//            centre of mass, in the floating-base frame (oMi[1].actInv(oYcrb[0]).inertia() after computeCentroidalMap)
//   hip_xy = origins of the four *_hip_joint's relative to the floating base (FL FR RL RR)
// pinocchio is a third-party dependency of the reference (absent here, unpinned there: SURVEY.md section 8c); this header is
// the facade's own small URDF reader: kinematic tree of revolute / fixed joints, parallel-axis sum.  Host code, no GPU.
#pragma once
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdlib>
#include <fstream>
#include <map>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "../srbm_rti.h"

namespace mpc {
namespace urdf_detail {

struct M3 {
    double m[3][3];
    static M3 I() { return {{{1, 0, 0}, {0, 1, 0}, {0, 0, 1}}}; }
    M3 operator*(const M3& o) const { M3 r{}; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { double a = 0; for (int k = 0; k < 3; k++) a += m[i][k] * o.m[k][j]; r.m[i][j] = a; } return r; }
    M3 T() const { M3 r{}; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r.m[i][j] = m[j][i]; return r; }
    std::array<double, 3> operator*(const std::array<double, 3>& v) const { return {m[0][0] * v[0] + m[0][1] * v[1] + m[0][2] * v[2], m[1][0] * v[0] + m[1][1] * v[1] + m[1][2] * v[2], m[2][0] * v[0] + m[2][1] * v[1] + m[2][2] * v[2]}; }
};
using V3 = std::array<double, 3>;
inline V3 add(const V3& a, const V3& b) { return {a[0] + b[0], a[1] + b[1], a[2] + b[2]}; }
inline V3 sub(const V3& a, const V3& b) { return {a[0] - b[0], a[1] - b[1], a[2] - b[2]}; }

inline M3 rpy(double r, double p, double y) {
    const double cr = std::cos(r), sr = std::sin(r), cp = std::cos(p), sp = std::sin(p), cy = std::cos(y), sy = std::sin(y);
    const M3 Rx{{{1, 0, 0}, {0, cr, -sr}, {0, sr, cr}}}, Ry{{{cp, 0, sp}, {0, 1, 0}, {-sp, 0, cp}}}, Rz{{{cy, -sy, 0}, {sy, cy, 0}, {0, 0, 1}}};
    return Rz * Ry * Rx;
}
inline M3 axis_angle(V3 a, double q) {
    const double n = std::sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
    for (double& x : a) x /= n;
    const M3 K{{{0, -a[2], a[1]}, {a[2], 0, -a[0]}, {-a[1], a[0], 0}}};
    const M3 K2 = K * K;
    M3 R = M3::I();
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) R.m[i][j] += std::sin(q) * K.m[i][j] + (1 - std::cos(q)) * K2.m[i][j];
    return R;
}
inline M3 quat_xyzw(const double* q) {
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    return {{{1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)}, {2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)},
             {2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)}}};
}

// ---- a tag scanner that is enough for URDF files: elements by name, attributes by name ----
inline std::string attr(const std::string& tag, const std::string& name, const std::string& dflt = "") {
    size_t p = 0;
    while ((p = tag.find(name + "=", p)) != std::string::npos) {
        if (p > 0 && (std::isalnum((unsigned char)tag[p - 1]) || tag[p - 1] == '_')) { p += name.size(); continue; }
        const size_t q0 = p + name.size() + 1;
        if (q0 >= tag.size()) break;
        const char quote = tag[q0];
        const size_t q1 = tag.find(quote, q0 + 1);
        if (q1 == std::string::npos) break;
        return tag.substr(q0 + 1, q1 - q0 - 1);
    }
    return dflt;
}
inline V3 vec3(const std::string& s, V3 d = {0, 0, 0}) {
    if (s.empty()) return d;
    std::istringstream is(s);
    V3 v{};
    is >> v[0] >> v[1] >> v[2];
    return v;
}
// the opening tags `<name ...>` found inside `text` (not nested deeper than their first closing tag), with their bodies
struct Element { std::string tag, body; };
inline std::vector<Element> elements(const std::string& text, const std::string& name) {
    std::vector<Element> out;
    size_t p = 0;
    const std::string open = "<" + name;
    while ((p = text.find(open, p)) != std::string::npos) {
        const char c = text[p + open.size()];
        if (!(c == ' ' || c == '>' || c == '/' || c == '\n' || c == '\t' || c == '\r')) { p += open.size(); continue; }
        const size_t e = text.find('>', p);
        if (e == std::string::npos) break;
        Element el;
        el.tag = text.substr(p, e - p + 1);
        if (text[e - 1] != '/') {
            const size_t c2 = text.find("</" + name + ">", e);
            if (c2 == std::string::npos) break;
            el.body = text.substr(e + 1, c2 - e - 1);
            p = c2;
        } else p = e;
        out.push_back(el);
    }
    return out;
}
inline std::string strip_comments(std::string s) {
    size_t p;
    while ((p = s.find("<!--")) != std::string::npos) {
        const size_t e = s.find("-->", p);
        s.erase(p, e == std::string::npos ? std::string::npos : e - p + 3);
    }
    return s;
}

struct Link { bool has = false; double mass = 0; V3 com{}; M3 Rl = M3::I(); M3 I{}; };
struct Joint { std::string name, type, parent, child; V3 xyz{}; M3 R = M3::I(); V3 axis{1, 0, 0}; };

}  // namespace urdf_detail

namespace urdf_detail {
struct Robot {
    std::map<std::string, Link> links;
    std::vector<Joint> joints;
    std::map<std::string, std::vector<const Joint*>> children;
    std::string root;
    const Joint* joint(const std::string& name) const { for (const Joint& j : joints) if (j.name == name) return &j; return nullptr; }
};
// links (inertial data) and joints (parent, child, origin, axis) of a URDF file; the root is the link that is a parent but nobody's child
inline void parse(const std::string& urdf_path, Robot& R) {
    std::ifstream f(urdf_path);
    if (!f) throw std::runtime_error("Could not open the URDF: " + urdf_path);
    std::stringstream ss;
    ss << f.rdbuf();
    const std::string text = strip_comments(ss.str());
    for (const Element& l : elements(text, "link")) {
        Link L;
        const auto inr = elements(l.body, "inertial");
        if (!inr.empty()) {
            L.has = true;
            const auto org = elements(inr[0].body, "origin");
            if (!org.empty()) { L.com = vec3(attr(org[0].tag, "xyz")); const V3 r = vec3(attr(org[0].tag, "rpy")); L.Rl = rpy(r[0], r[1], r[2]); }
            L.mass = std::atof(attr(elements(inr[0].body, "mass").at(0).tag, "value", "0").c_str());
            const std::string it = elements(inr[0].body, "inertia").at(0).tag;
            auto g = [&](const char* k) { return std::atof(attr(it, k, "0").c_str()); };
            L.I = {{{g("ixx"), g("ixy"), g("ixz")}, {g("ixy"), g("iyy"), g("iyz")}, {g("ixz"), g("iyz"), g("izz")}}};
        }
        R.links[attr(l.tag, "name")] = L;
    }
    for (const Element& j : elements(text, "joint")) {
        const auto par = elements(j.body, "parent");
        if (par.empty()) continue;                               // <transmission> blocks reuse the <joint> tag
        Joint J;
        J.name = attr(j.tag, "name"); J.type = attr(j.tag, "type");
        J.parent = attr(par[0].tag, "link"); J.child = attr(elements(j.body, "child").at(0).tag, "link");
        const auto org = elements(j.body, "origin");
        if (!org.empty()) { J.xyz = vec3(attr(org[0].tag, "xyz")); const V3 r = vec3(attr(org[0].tag, "rpy")); J.R = rpy(r[0], r[1], r[2]); }
        const auto ax = elements(j.body, "axis");
        if (!ax.empty()) J.axis = vec3(attr(ax[0].tag, "xyz"), {1, 0, 0});
        R.joints.push_back(J);
    }
    if (R.links.empty() || R.joints.empty()) throw std::runtime_error("No links / joints found in the URDF: " + urdf_path);
    std::map<std::string, bool> is_child;
    for (const Joint& j : R.joints) { R.children[j.parent].push_back(&j); is_child[j.child] = true; }
    for (const Joint& j : R.joints) if (!is_child.count(j.parent)) { R.root = j.parent; break; }
    if (R.root.empty()) throw std::runtime_error("The URDF has no root link: " + urdf_path);
}
}  // namespace urdf_detail

// nominal configuration: [base position (3), base quaternion xyzw (4), joint angles in pinocchio's model order]
inline srbm_model ModelConstantsFromUrdf(const std::string& urdf_path, const std::vector<double>& nom_config) {
    using namespace urdf_detail;
    Robot robot;
    parse(urdf_path, robot);
    std::map<std::string, Link>& links = robot.links;
    std::vector<Joint>& joints = robot.joints;
    std::map<std::string, std::vector<const Joint*>>& children = robot.children;
    const std::string root = robot.root;
    (void)joints;
    // actuated joints in pinocchio's model order: depth first, children in alphabetical order of the child link
    std::vector<std::string> order;
    struct Rec { static void visit(const std::string& link, std::map<std::string, std::vector<const Joint*>>& ch, std::vector<std::string>& ord) {
        auto v = ch[link];
        std::sort(v.begin(), v.end(), [](const Joint* a, const Joint* b) { return a->child < b->child; });
        for (const Joint* j : v) { if (j->type == "revolute" || j->type == "continuous") ord.push_back(j->name); visit(j->child, ch, ord); }
    } };
    Rec::visit(root, children, order);
    if (nom_config.size() < 7 + order.size()) throw std::runtime_error("The nominal configuration is shorter than 7 + the number of actuated joints.");
    std::map<std::string, double> q;
    for (size_t k = 0; k < order.size(); k++) q[order[k]] = nom_config[7 + k];
    const V3 base_p{nom_config[0], nom_config[1], nom_config[2]};
    const M3 base_R = quat_xyzw(&nom_config[3]);
    struct Body { double m; V3 c; M3 I; };
    std::vector<Body> bodies;
    std::map<std::string, V3> hips;
    struct Walk { static void go(const std::string& link, const V3& p, const M3& R, std::map<std::string, Link>& links,
                                 std::map<std::string, std::vector<const Joint*>>& ch, std::map<std::string, double>& q, std::vector<Body>& bodies,
                                 std::map<std::string, V3>& hips, const V3& base_p, const M3& base_R) {
        const Link& L = links[link];
        if (L.has) bodies.push_back({L.mass, add(p, R * L.com), R * L.Rl * L.I * L.Rl.T() * R.T()});
        for (const Joint* j : ch[link]) {
            const V3 pj = add(p, R * j->xyz);
            M3 Rj = R * j->R;
            const std::string suffix = "_hip_joint";
            if (j->name.size() > suffix.size() && j->name.compare(j->name.size() - suffix.size(), suffix.size(), suffix) == 0)
                hips[j->name.substr(0, 2)] = base_R.T() * sub(pj, base_p);
            if (j->type == "revolute" || j->type == "continuous") Rj = Rj * axis_angle(j->axis, q[j->name]);
            go(j->child, pj, Rj, links, ch, q, bodies, hips, base_p, base_R);
        }
    } };
    Walk::go(root, base_p, base_R, links, children, q, bodies, hips, base_p, base_R);
    srbm_model out{};
    double mass = 0;
    V3 com{0, 0, 0};
    for (const Body& b : bodies) { mass += b.m; for (int i = 0; i < 3; i++) com[i] += b.m * b.c[i]; }
    for (double& x : com) x /= mass;
    M3 Iw{};
    for (const Body& b : bodies) {
        const V3 d = sub(b.c, com);
        const double dd = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Iw.m[i][j] += b.I.m[i][j] + b.m * ((i == j ? dd : 0.0) - d[i] * d[j]);
    }
    const M3 Ir = base_R.T() * Iw * base_R;
    out.mass = mass;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) out.Ir[3 * i + j] = Ir.m[i][j];
    const char* names[4] = {"FL", "FR", "RL", "RR"};
    for (int e = 0; e < 4; e++) {
        if (!hips.count(names[e])) throw std::runtime_error(std::string("No ") + names[e] + "_hip_joint in the URDF.");
        out.hip_xy[2 * e] = hips[names[e]][0]; out.hip_xy[2 * e + 1] = hips[names[e]][1];
    }
    return out;
}


// Leg geometry for the kinematics entries (srbm_set_leg_kinematics): per leg FL FR RL RR the origins of the hip joint in the trunk, the
// thigh joint in the hip, the calf joint in the thigh and the foot frame in the calf (what pinocchio builds its joint placements from).
// The closed-form kinematics of the library are written for joint axes x, y, y and origins without rotation: anything else is refused.
inline srbm_leg_kinematics LegKinematicsFromUrdf(const std::string& urdf_path) {
    using namespace urdf_detail;
    Robot robot;
    parse(urdf_path, robot);
    srbm_leg_kinematics out{};
    const char* legs[4] = {"FL", "FR", "RL", "RR"};
    const char* suffix[4] = {"_hip_joint", "_thigh_joint", "_calf_joint", "_foot_fixed"};
    for (int e = 0; e < 4; e++)
        for (int k = 0; k < 4; k++) {
            const Joint* j = robot.joint(std::string(legs[e]) + suffix[k]);
            if (!j) throw std::runtime_error(std::string("No ") + legs[e] + suffix[k] + " in the URDF.");
            for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++)
                if (std::fabs(j->R.m[a][b] - (a == b ? 1.0 : 0.0)) > 1e-12) throw std::runtime_error("Joint origin with a rotation: " + j->name);
            if (k < 3) {
                const V3 want = k == 0 ? V3{1, 0, 0} : V3{0, 1, 0};
                for (int a = 0; a < 3; a++) if (std::fabs(j->axis[a] - want[a]) > 1e-12) throw std::runtime_error("Unexpected joint axis: " + j->name);
            }
            for (int a = 0; a < 3; a++) out.origin[e][k][a] = j->xyz[a];
        }
    return out;
}

// Rigid bodies of the whole-body model (srbm_set_wbc_model: body_mass / body_com / body_inertia): the trunk and the three moving links of
// every leg, every link hanging on a FIXED joint merged into its parent (as pinocchio's URDF parser does): mass, centre of mass and
// rotational inertia about it, in the frame of the body's joint (trunk: the floating-base frame).  Gains and weights are not touched.
inline void WbcBodiesFromUrdf(const std::string& urdf_path, srbm_wbc_model* out) {
    using namespace urdf_detail;
    Robot robot;
    parse(urdf_path, robot);
    struct Part { double m; V3 c; M3 I; };
    struct Merge { static void go(const Robot& R, const std::string& link, const V3& p, const M3& Rot, std::vector<Part>& parts) {
        const auto it = R.links.find(link);
        if (it != R.links.end() && it->second.has) { const Link& L = it->second; parts.push_back({L.mass, add(p, Rot * L.com), Rot * L.Rl * L.I * L.Rl.T() * Rot.T()}); }
        const auto ch = R.children.find(link);
        if (ch == R.children.end()) return;
        for (const Joint* j : ch->second) if (j->type == "fixed") go(R, j->child, add(p, Rot * j->xyz), Rot * j->R, parts);
    } };
    auto lump = [&](const std::string& link, int b) {
        std::vector<Part> parts;
        Merge::go(robot, link, V3{0, 0, 0}, M3::I(), parts);
        if (parts.empty()) throw std::runtime_error("Link without inertial data: " + link);
        double m = 0; V3 c{0, 0, 0};
        for (const Part& q : parts) { m += q.m; for (int i = 0; i < 3; i++) c[i] += q.m * q.c[i]; }
        for (double& x : c) x /= m;
        M3 I{};
        for (const Part& q : parts) {
            const V3 d = sub(q.c, c);
            const double dd = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
            for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) I.m[i][j] += q.I.m[i][j] + q.m * ((i == j ? dd : 0.0) - d[i] * d[j]);
        }
        out->body_mass[b] = m;
        for (int i = 0; i < 3; i++) out->body_com[b][i] = c[i];
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) out->body_inertia[b][3 * i + j] = I.m[i][j];
    };
    const Joint* fl = robot.joint("FL_hip_joint");
    if (!fl) throw std::runtime_error("No FL_hip_joint in the URDF.");
    lump(fl->parent, 0);                                         // the trunk: the link the hips hang on
    const char* legs[4] = {"FL", "FR", "RL", "RR"};
    const char* part[3] = {"_hip_joint", "_thigh_joint", "_calf_joint"};
    for (int e = 0; e < 4; e++)
        for (int k = 0; k < 3; k++) {
            const Joint* j = robot.joint(std::string(legs[e]) + part[k]);
            if (!j) throw std::runtime_error(std::string("No ") + legs[e] + part[k] + " in the URDF.");
            lump(j->child, 1 + 3 * e + k);
        }
}

}  // namespace mpc
