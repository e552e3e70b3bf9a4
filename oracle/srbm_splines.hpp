// ORACLE (test infrastructure, NOT product code).
// CPU restatement of the reference's per-foot phase-based Hermite splines.
// Follows /root/reference/mpc/spline/end_effector_splines.cpp (cited per function) and
// /root/reference/mpc/spline/spline_node.cpp.  Only tests/, __graft_entry__.smoke() and
// bench.py's cpu_baseline leg may use anything under oracle/.
//
// Compile with -ffp-contract=off: node lookups compare doubles produced by exactly the
// operation order used in the reference (SURVEY.md §7 "FP-sensitive integer decisions").
#pragma once
#include <array>
#include <cassert>
#include <cmath>
#include <stdexcept>
#include <utility>
#include <vector>

namespace orc {

enum TimeType { LiftOff = 0, TouchDown = 1, Inter = 2 };      // end_effector_splines.h:11-15
enum NodeType { NoDeriv = 0, FullDeriv = 1, Empty = 2 };      // spline_node.h:14-18
enum SplineType { Force = 0, Position = 1 };                   // end_effector_splines.h:38-41

struct SplineTime {                                            // end_effector_splines.h:17-31
    double time = -1;
    TimeType type = Inter;
};
using time_v = std::vector<SplineTime>;

struct SplineNode {                                            // spline_node.cpp:8-40
    NodeType type;
    double v0 = 0, v1 = 0;
    void SetVars(double a, double b) {
        if (type == Empty) throw std::runtime_error("Can't set the vars in this node. This node is set to empty.");
        if (type == NoDeriv) { v0 = a; } else { v0 = a; v1 = b; }
    }
    double Get0() const {
        if (type == Empty) throw std::runtime_error("Can't get the vars in this node. This node is set to empty.");
        return v0;
    }
    double Get1() const {
        if (type == Empty) throw std::runtime_error("Can't get the vars in this node. This node is set to empty.");
        return v1;
    }
};

class EndEffectorSplines {
public:
    static constexpr double FORCE_MULT = 100;                  // end_effector_splines.h:152
    static constexpr int POS_VARS = 3;
    using node_v = std::vector<SplineNode>;

    // end_effector_splines.cpp:34-153
    EndEffectorSplines(int num_contacts, const std::vector<double>& times, bool start_in_contact,
                       int num_force_polys) : num_force_polys_(num_force_polys) {
        if (num_force_polys_ < 2)
            throw std::runtime_error("The number of force polynomials between constant sections must be at least 2.");
        int num_times_in_pattern = num_force_polys + 1;
        spline_stride_ = num_force_polys;
        if (num_force_polys % 2) num_times_in_pattern++;
        for (int i = 0; i < num_times_in_pattern; i++) {
            if (!start_in_contact) {
                if (i == 0)      push_pattern(NoDeriv, NoDeriv, NoDeriv, LiftOff);
                else if (i == 1) push_pattern(Empty, Empty, FullDeriv, Inter);
                else if (i == 2) push_pattern(NoDeriv, NoDeriv, NoDeriv, TouchDown);
                else             push_pattern(FullDeriv, Empty, Empty, Inter);
            } else {
                if (i == 0)                              push_pattern(NoDeriv, NoDeriv, NoDeriv, TouchDown);
                else if (i < num_force_polys)            push_pattern(FullDeriv, Empty, Empty, Inter);
                else if (i == num_times_in_pattern - 1)  push_pattern(Empty, Empty, FullDeriv, Inter);
                else                                     push_pattern(NoDeriv, NoDeriv, NoDeriv, LiftOff);
            }
        }
        const int P = (int)force_pat_.size();
        for (int coord = 0; coord < POS_VARS; coord++) {
            int i = 0, j = 0, k = 1;
            if (coord < 2) {
                while (i < num_contacts) {
                    forces_[coord].push_back({force_pat_[j % P]});
                    positions_[coord].push_back({pos_pat_[j % P]});
                    if (force_pat_[j % P] == FullDeriv) {
                        if (coord == 0) {
                            // :115-117  t_{i-1} + k*(t_i - t_{i-1})/nfp   (operation order kept)
                            times_.push_back({times.at(i - 1) + k * (times.at(i) - times.at(i - 1)) / (num_force_polys),
                                              time_pat_[j % P]});
                            k++;
                        }
                    } else if (force_pat_[j % P] == Empty) {
                        if (coord == 0)
                            times_.push_back({times.at(i - 1) + (times.at(i) - times.at(i - 1)) / 2, time_pat_[j % P]});
                    } else {
                        if (coord == 0) times_.push_back({times.at(i), time_pat_[j % P]});
                        i++;
                        k = 1;
                    }
                    j++;
                }
            } else {
                while (i < num_contacts) {
                    forces_[coord].push_back({force_pat_[j % P]});
                    positions_[coord].push_back({zpos_pat_[j % P]});
                    if (!(force_pat_[j % P] == FullDeriv || force_pat_[j % P] == Empty)) i++;
                    j++;
                }
            }
        }
        assert(forces_[1].size() == positions_[1].size());
        assert(forces_[1].size() == times_.size());
    }

    // :169-199
    double ValueAt(SplineType type, int coord, double time) const {
        const node_v& spline = Select(type, coord);
        const int lower_node = GetLowerNodeIdx(type, coord, time);
        const int upper_node = GetUpperNodeIdx(type, coord, time);
        if (upper_node == lower_node) return spline.at(lower_node).Get0();
        const double deltat = times_.at(upper_node).time - times_.at(lower_node).time;
        const double time_spline = time - times_.at(lower_node).time;
        double v0 = spline.at(lower_node).Get0();
        double v1 = spline.at(upper_node).Get0();
        double v2 = spline.at(lower_node).Get1();
        double v3 = spline.at(upper_node).Get1();
        if (type == Force) { v2 *= FORCE_MULT; v3 *= FORCE_MULT; }
        const double a2 = -(1 / std::pow(deltat, 2)) * 3 * (v0 - v1) - (1 / deltat) * (2 * v2 + v3);
        const double a3 = (1 / std::pow(deltat, 3)) * 2 * (v0 - v1) + (1 / std::pow(deltat, 2)) * (v2 + v3);
        return v0 + v2 * time_spline + a2 * std::pow(time_spline, 2) + a3 * std::pow(time_spline, 3);
    }

    // :201-282
    std::vector<double> GetPolyVarsLin(SplineType type, int coord, double time) const {
        const node_v& spline = Select(type, coord);
        const int lower_node = GetLowerNodeIdx(type, coord, time);
        const int upper_node = GetUpperNodeIdx(type, coord, time);
        if (lower_node == upper_node) return {1.0};
        const double poly_time = time - times_.at(lower_node).time;
        const double deltat = times_.at(upper_node).time - times_.at(lower_node).time;
        if (type == Force) {
            const NodeType lt = spline.at(lower_node).type, ut = spline.at(upper_node).type;
            if (lt == NoDeriv && ut == NoDeriv)
                throw std::runtime_error("There is no mutable variables at the provided time.");
            if (lt == NoDeriv && ut == FullDeriv)
                return {x1Coef(poly_time, deltat), x1dotCoef(poly_time, deltat) * FORCE_MULT};
            if (lt == FullDeriv && ut == NoDeriv)
                return {x0Coef(poly_time, deltat), x0dotCoef(poly_time, deltat) * FORCE_MULT};
            return {x0Coef(poly_time, deltat), x0dotCoef(poly_time, deltat) * FORCE_MULT,
                    x1Coef(poly_time, deltat), x1dotCoef(poly_time, deltat) * FORCE_MULT};
        }
        if (coord != 2) {
            if (forces_[coord].at(lower_node).type == NoDeriv && forces_[coord].at(lower_node + 2).type == NoDeriv)
                return {x0Coef(poly_time, deltat), x1Coef(poly_time, deltat)};
            return {1.0};
        }
        const NodeType lt = positions_[coord].at(lower_node).type, ut = positions_[coord].at(upper_node).type;
        if (lt == NoDeriv && ut == FullDeriv)
            return {x0Coef(poly_time, deltat), x1Coef(poly_time, deltat), x1dotCoef(poly_time, deltat)};
        if (lt == FullDeriv && ut == NoDeriv)
            return {x0Coef(poly_time, deltat), x0dotCoef(poly_time, deltat), x1Coef(poly_time, deltat)};
        return {1.0};
    }

    // :284-354
    std::pair<int, int> GetVarsIdx(SplineType type, int coord, double time) const {
        const node_v& spline = Select(type, coord);
        const int lower_node = GetLowerNodeIdx(type, coord, time);
        const int upper_node = GetUpperNodeIdx(type, coord, time);
        const std::vector<int> mut_nodes = GetMutableNodes(type, coord);
        int vars_idx = 0;
        if (type == Force) {
            for (int i = 0; i < (int)mut_nodes.size(); i++)
                if (mut_nodes[i] < lower_node) vars_idx = 2 * (i + 1);
            const NodeType lt = spline.at(lower_node).type, ut = spline.at(upper_node).type;
            if (lt == NoDeriv && ut == NoDeriv)
                throw std::runtime_error("There is no mutable variables at the provided time.");
            if ((lt == NoDeriv && ut == FullDeriv) || (lt == FullDeriv && ut == NoDeriv)) return {vars_idx, 2};
            if (lower_node == upper_node) return {vars_idx, 1};
            return {vars_idx, 4};
        }
        vars_idx--;
        for (int i = 0; i < (int)mut_nodes.size(); i++)
            if (mut_nodes[i] <= lower_node) vars_idx++;
        if (lower_node == upper_node) return {vars_idx, 1};
        if (coord != 2) {
            if (forces_[coord].at(lower_node).type == NoDeriv && forces_[coord].at(lower_node + 2).type == NoDeriv)
                return {vars_idx, 2};
            return {vars_idx, 1};
        }
        for (int i = 0; i < (int)mut_nodes.size(); i++)
            if (positions_[coord].at(mut_nodes[i]).type == FullDeriv && mut_nodes[i] < lower_node) vars_idx++;
        const NodeType lt = positions_[coord].at(lower_node).type, ut = positions_[coord].at(upper_node).type;
        if ((lt == NoDeriv && ut == FullDeriv) || (lt == FullDeriv && ut == NoDeriv)) return {vars_idx, 3};
        if (forces_[coord].at(lower_node).type == NoDeriv && forces_[coord].at(lower_node + 2).type == NoDeriv)
            return {vars_idx, 2};
        return {vars_idx, 1};
    }

    // :356-364
    bool IsForceMutable(double time) const {
        const int lower_node = GetLowerNodeIdx(Force, 0, time);
        const int upper_node = GetUpperNodeIdx(Force, 0, time);
        return !(forces_[0].at(lower_node).type == NoDeriv && forces_[0].at(upper_node).type == NoDeriv);
    }

    // :366-449
    void AddPoly(double additional_time) {
        const int num_nodes = GetNumNodes();
        if (forces_[0].at(num_nodes - 1).type == NoDeriv && forces_[0].at(num_nodes - 2).type == FullDeriv) {
            // last phase was a stance -> append a swing: mid-swing node then touchdown
            for (int i = 0; i < 2; i++) {
                for (int coord = 0; coord < POS_VARS; coord++) {
                    if (i == 0) {
                        forces_[coord].push_back({Empty});
                        if (coord == 0) times_.push_back({times_.back().time + additional_time / 2, Inter});
                        positions_[coord].push_back({coord == 2 ? FullDeriv : Empty});
                    } else {
                        forces_[coord].push_back({NoDeriv});
                        positions_[coord].push_back({NoDeriv});
                        if (coord == 0) times_.push_back({times_.back().time + additional_time / 2, TouchDown});
                    }
                }
            }
        } else {
            // last phase was a swing -> append a stance: (nfp-1) interior force nodes then lift-off
            for (int coord = 0; coord < POS_VARS; coord++) {
                for (int i = 0; i < num_force_polys_ - 1; i++) {
                    forces_[coord].push_back({FullDeriv});
                    positions_[coord].push_back({Empty});
                    if (coord == 0) times_.push_back({times_.back().time + additional_time / num_force_polys_, Inter});
                }
                forces_[coord].push_back({NoDeriv});
                positions_[coord].push_back({NoDeriv});
                if (coord == 0) times_.push_back({times_.back().time + additional_time / num_force_polys_, LiftOff});
            }
        }
    }

    // :451-465
    void RemovePoly(double start_time) {
        const int lower_node = GetLowerNodeIdx(Position, 0, start_time);
        if (lower_node != 0) {
            times_.erase(times_.begin(), times_.begin() + lower_node);
            for (int coord = 0; coord < POS_VARS; coord++) {
                forces_[coord].erase(forces_[coord].begin(), forces_[coord].begin() + lower_node);
                positions_[coord].erase(positions_[coord].begin(), positions_[coord].begin() + lower_node);
            }
        }
        if (GetLowerNodeIdx(Position, 0, start_time) != 0) throw std::runtime_error("Poly remove did not work.");
    }

    // :513-648
    double ComputePartialWrtTime(SplineType type, int coord, double time, int time_idx) const {
        const node_v& spline = Select(type, coord);
        const int upper_node = GetUpperNodeIdx(type, coord, time);
        const int lower_node = GetLowerNodeIdx(type, coord, time);
        const double deltat = times_.at(upper_node).time - times_.at(lower_node).time;
        const double ts = time - times_.at(lower_node).time;
        const int node = ConvertContactNodeToSplineNode(time_idx);
        const bool direct_dep = (node == lower_node || node == upper_node);
        const bool wrt_lower = (node == lower_node);
        const double x0 = spline.at(lower_node).Get0();
        const double x1 = spline.at(upper_node).Get0();
        double x0dot = 0, x1dot = 0;
        if (spline.at(lower_node).type == FullDeriv)
            x0dot = (type == Force) ? spline.at(lower_node).v1 * FORCE_MULT : spline.at(lower_node).v1;
        if (spline.at(upper_node).type == FullDeriv)
            x1dot = (type == Force) ? spline.at(upper_node).v1 * FORCE_MULT : spline.at(upper_node).v1;
        const double nfp = static_cast<double>(num_force_polys_);

        auto da2 = [&](double dD) {
            return 6 * std::pow(deltat, -3) * (x0 - x1) * dD + (2 * x0dot + x1dot) * std::pow(deltat, -2) * dD;
        };
        auto da3 = [&](double dD) {
            return -6 * std::pow(deltat, -4) * (x0 - x1) * dD - 2 * std::pow(deltat, -3) * (x0dot + x1dot) * dD;
        };
        const double a2 = -std::pow(deltat, -2) * (3 * (x0 - x1) + deltat * (2 * x0dot + x1dot));
        const double a3 = std::pow(deltat, -3) * (2 * (x0 - x1) + deltat * (x0dot + x1dot));

        if (direct_dep && wrt_lower) {                                        // :554-576
            double dDTdth = -1.0;
            if (type == Force) dDTdth = -1.0 / nfp;
            return da2(dDTdth) * std::pow(ts, 2) + da3(dDTdth) * std::pow(ts, 3) - x0dot - a2 * 2 * ts -
                   a3 * 3 * std::pow(ts, 2);
        }
        if (direct_dep && !wrt_lower) {                                       // :578-595
            double dDTdth = 1.0, dtdth = 0.0;
            if (type == Force) {
                dDTdth = 1.0 / nfp;
                dtdth = -static_cast<double>(num_force_polys_ - 1) / nfp;
            }
            return da2(dDTdth) * std::pow(ts, 2) + da3(dDTdth) * std::pow(ts, 3) +
                   (x0dot + a2 * 2 * ts + a3 * 3 * std::pow(ts, 2)) * dtdth;
        }
        if (node > upper_node && node <= GetUpperNodeIdx(Position, 0, time)) {   // :598-620
            const double dDTdth = 1.0 / nfp;
            const int j = CountFullDerivBack(coord, lower_node);
            const double dtdth = -static_cast<double>(j) / nfp;
            return da2(dDTdth) * std::pow(ts, 2) + da3(dDTdth) * std::pow(ts, 3) +
                   (x0dot + a2 * 2 * ts + a3 * 3 * std::pow(ts, 2)) * dtdth;
        }
        if (node < lower_node && node >= GetLowerNodeIdx(Position, 0, time)) {   // :622-645
            const double dDTdth = -1.0 / nfp;
            const int j = CountFullDerivBack(coord, lower_node);
            const double dtdth = -(static_cast<double>(-j) / nfp + 1.0);
            return da2(dDTdth) * std::pow(ts, 2) + da3(dDTdth) * std::pow(ts, 3) +
                   (x0dot + a2 * 2 * ts + a3 * 3 * std::pow(ts, 2)) * dtdth;
        }
        return 0;
    }

    // :650-803
    std::vector<double> ComputeCoefPartialWrtTime(SplineType type, int coord, double time, int time_idx,
                                                  double dtwdth = 0) const {
        const node_v& spline = Select(type, coord);
        const int upper_node = GetUpperNodeIdx(type, coord, time);
        const int lower_node = GetLowerNodeIdx(type, coord, time);
        double deltat = times_.at(upper_node).time - times_.at(lower_node).time;
        if (deltat == 0)
            deltat = times_.at(upper_node).time - times_.at(GetLowerNodeIdx(type, coord, time - 1e-4)).time;
        const double ts = time - times_.at(lower_node).time;
        int vars_index, vars_affecting;
        std::tie(vars_index, vars_affecting) = GetVarsIdx(type, coord, time);
        (void)vars_index;
        std::vector<double> cp(vars_affecting, 0.0);
        const int node = ConvertContactNodeToSplineNode(time_idx);
        const bool direct_dep = (node == lower_node || node == upper_node);
        bool wrt_lower = (node == lower_node);
        double dtdth = dtwdth;
        const double nfp = static_cast<double>(num_force_polys_);

        if (type == Force) {
            const int j = CountFullDerivBack(coord, lower_node);
            double dDTdth = 1.0 / nfp;
            if (wrt_lower) {
                dDTdth = -1.0 / nfp;
                dtdth += static_cast<double>(j) / nfp - 1.0;
            } else {
                dtdth += -static_cast<double>(j) / nfp;
            }
            if (direct_dep) {
                if (wrt_lower) {
                    cp.at(0) = x1CoefPartial(ts, deltat, dtdth, dDTdth);
                    cp.at(1) = x1dotCoefPartial(ts, deltat, dtdth, dDTdth) * FORCE_MULT;
                } else {
                    cp.at(0) = x0CoefPartial(ts, deltat, dtdth, dDTdth);
                    cp.at(1) = x0dotCoefPartial(ts, deltat, dtdth, dDTdth) * FORCE_MULT;
                }
            } else {
                bool active = false;
                if (node > upper_node && node <= GetUpperNodeIdx(Position, 0, time)) {
                    wrt_lower = false;
                    dDTdth = 1.0 / nfp;
                    dtdth = dtwdth - static_cast<double>(j) / nfp;
                    active = true;
                } else if (node < lower_node && node >= GetLowerNodeIdx(Position, 0, time)) {
                    wrt_lower = true;
                    dDTdth = -1.0 / nfp;
                    dtdth = dtwdth + static_cast<double>(j) / nfp - 1.0;
                    active = true;
                }
                if (active) {
                    if (spline.at(lower_node).type == FullDeriv) {
                        cp.at(0) = x0CoefPartial(ts, deltat, dtdth, dDTdth);
                        cp.at(1) = FORCE_MULT * x0dotCoefPartial(ts, deltat, dtdth, dDTdth);
                        if (spline.at(upper_node).type == FullDeriv) {
                            cp.at(2) = x1CoefPartial(ts, deltat, dtdth, dDTdth);
                            cp.at(3) = FORCE_MULT * x1dotCoefPartial(ts, deltat, dtdth, dDTdth);
                        }
                    } else if (spline.at(upper_node).type == FullDeriv) {
                        cp.at(0) = x1CoefPartial(ts, deltat, dtdth, dDTdth);
                        cp.at(1) = FORCE_MULT * x1dotCoefPartial(ts, deltat, dtdth, dDTdth);
                    }
                }
            }
        } else {
            double dDTdth = 1.0;
            if (wrt_lower) { dtdth += -1.0; dDTdth = -1.0; }
            if (direct_dep) {
                if (lower_node == upper_node) {
                    cp.at(0) = 0;
                } else if (positions_[2].at(lower_node + 1).type == FullDeriv) {
                    cp.at(0) = x0CoefPartial(ts, deltat, dtdth, dDTdth);
                    cp.at(1) = x1CoefPartial(ts, deltat, dtdth, dDTdth);
                } else {
                    cp.at(0) = 0;
                }
            }
        }
        return cp;
    }

    // :805-813
    bool IsInContact(double time) const {
        const int lower_node = GetLowerNodeIdx(Position, 0, time);
        const int upper_node = GetUpperNodeIdx(Position, 0, time);
        return times_.at(lower_node).type == TouchDown && times_.at(upper_node).type == LiftOff;
    }

    // :815-858
    void SetVars(SplineType type, int coord, int node_idx, double a, double b) {
        node_v& spline = Select(type, coord);
        if (spline.at(node_idx).type == Empty)
            throw std::runtime_error("Can't set this node's variables. This node is empty.");
        if (type == Force && spline.at(node_idx).type == NoDeriv)
            throw std::runtime_error("Force spline cannot be changed at that node. Always set to 0.");
        const int sz = (int)spline.size();
        if (type == Position && coord != 2) {
            if (node_idx < sz - 1 && forces_[coord].at(node_idx + 1).type == FullDeriv) {
                spline.at(node_idx).SetVars(a, b);
                spline.at(node_idx + spline_stride_).SetVars(a, b);
            } else if (node_idx > 0 && forces_[coord].at(node_idx - 1).type == FullDeriv) {
                spline.at(node_idx).SetVars(a, b);
                if (node_idx >= spline_stride_) spline.at(node_idx - spline_stride_).SetVars(a, b);
            } else {
                spline.at(node_idx).SetVars(a, b);
            }
        } else if (type == Position) {
            if (node_idx < sz - 1 && positions_[coord].at(node_idx).type != FullDeriv &&
                forces_[coord].at(node_idx + 1).type == FullDeriv) {
                spline.at(node_idx).SetVars(a, b);
                spline.at(node_idx + spline_stride_).SetVars(a, b);
            } else if (node_idx > 0 && positions_[coord].at(node_idx).type != FullDeriv &&
                       forces_[coord].at(node_idx - 1).type == FullDeriv) {
                spline.at(node_idx).SetVars(a, b);
                if (node_idx >= spline_stride_) spline.at(node_idx - spline_stride_).SetVars(a, b);
            } else {
                spline.at(node_idx).SetVars(a, b);
            }
        } else {
            spline.at(node_idx).SetVars(a, b);
        }
    }

    // :860-892
    void SetContactTimes(time_v& contact_times) {
        for (auto& ct : contact_times) {
            if (ct.time < 0 && std::abs(ct.time) < 1e-3) ct.time = 0;
            else if (ct.time < 0) throw std::runtime_error("Invalid time: negative");
        }
        int contact_idx = 0;
        for (int i = 0; i < GetNumNodes(); i++) {
            if (times_.at(i).type == LiftOff || times_.at(i).type == TouchDown) {
                times_.at(i).time = contact_times.at(contact_idx).time;
                contact_idx++;
            } else if (forces_[0].at(i).type == Empty) {
                times_.at(i).time = times_.at(i - 1).time +
                                    (contact_times.at(contact_idx).time - contact_times.at(contact_idx - 1).time) / 2;
            } else {
                double contact_time = 0.2 + contact_times.at(contact_idx - 1).time;
                if (contact_idx < (int)contact_times.size())
                    contact_time = contact_times.at(contact_idx).time - contact_times.at(contact_idx - 1).time;
                times_.at(i).time = times_.at(i - 1).time + contact_time / num_force_polys_;
            }
        }
    }

    NodeType GetNodeType(SplineType type, int coord, int node_idx) const { return Select(type, coord).at(node_idx).type; }
    int GetNumNodes() const { return (int)times_.size(); }

    // :905-940
    std::vector<int> GetMutableNodes(SplineType type, int coord) const {
        std::vector<int> mn;
        if (type == Force) {
            for (int i = 0; i < GetNumNodes(); i++)
                if (forces_[coord].at(i).type == FullDeriv) mn.push_back(i);
            return mn;
        }
        for (int i = 0; i < GetNumNodes(); i++) {
            if (positions_[coord].at(i).type != Empty) {
                mn.push_back(i);
                if (i + spline_stride_ < GetNumNodes() && positions_[coord].at(i + spline_stride_).type == NoDeriv)
                    i += spline_stride_;
            }
        }
        return mn;
    }

    std::vector<double> GetTimes() const {
        std::vector<double> t;
        for (auto& x : times_) t.push_back(x.time);
        return t;
    }

    // :950-979
    std::vector<double> GetSplineAsQPVec(SplineType type, int coord) const {
        const node_v& spline = Select(type, coord);
        std::vector<double> v;
        for (int it : GetMutableNodes(type, coord)) {
            if (spline.at(it).type == NoDeriv) v.push_back(spline.at(it).Get0());
            else { v.push_back(spline.at(it).Get0()); v.push_back(spline.at(it).Get1()); }
        }
        return v;
    }

    double GetEndTime() const { return times_.back().time; }
    double GetStartTime() const { return times_.front().time; }

    // :990-997
    int GetTotalPolyVars(SplineType type, int coord) const {
        return type == Force ? 2 * (int)GetMutableNodes(type, coord).size() : (int)GetMutableNodes(type, coord).size();
    }

    int GetNumContacts() const {
        int c = 0;
        for (auto& t : times_) if (t.type == LiftOff || t.type == TouchDown) c++;
        return c;
    }
    time_v GetContactTimes() const {                                      // :1022-1031
        time_v tv;
        for (auto& t : times_) if (t.type == LiftOff || t.type == TouchDown) tv.push_back(t);
        return tv;
    }

    // :1033-1040
    double GetNextTouchDownTime(double time) const {
        const int upper_node = GetUpperNodeIdx(Position, 0, time);
        if (times_.at(upper_node).type == TouchDown) return times_.at(upper_node).time;
        return times_.at(GetUpperNodeIdx(Position, 0, times_.at(upper_node).time + 0.001)).time;
    }

    // :1042-1060
    void SetToTouchdown(double time) {
        const int upper_node = GetUpperNodeIdx(Position, 0, time);
        if (times_.at(upper_node).type != TouchDown)
            throw std::runtime_error("Attempting to change a lift off to a touchdown node.");
        if (std::abs(times_.at(upper_node).time - time) > 1e-1)
            throw std::runtime_error("Attempting to change a touchdown node too far away from the current time.");
        const int upper_node2 = GetUpperNodeIdx(Position, 0, times_.at(upper_node).time + 0.001);
        const double time2 = times_.at(upper_node2).time;
        times_.at(upper_node).time = time;
        for (int i = 1; i < num_force_polys_; i++)
            times_.at(upper_node + i).time = i * (time2 - time) / num_force_polys_ + time;
    }

    // :1155-1163
    double GetSwingTime(double time) const {
        const int lower_node = GetLowerNodeIdx(Position, 0, time);
        if (times_.at(lower_node).type != LiftOff) return -1;
        const int upper_node = GetUpperNodeIdx(Position, 0, time);
        return times_.at(upper_node).time - times_.at(lower_node).time;
    }
    double GetFirstTDTime() const {
        for (auto& t : times_) if (t.type == TouchDown) return t.time;
        return 1e30;
    }

    // :1062-1084 (note the 1e4 upper clamp as coded)
    int GetLowerNodeIdx(SplineType type, int coord, double time) const {
        time = ClampTime(time);
        const node_v& spline = Select(type, coord);
        for (int i = (int)times_.size() - 1; i >= 0; i--)
            if (time >= times_.at(i).time && spline.at(i).type != Empty) return i;
        throw std::runtime_error("Invalid time.");
    }
    // :1086-1112
    int GetUpperNodeIdx(SplineType type, int coord, double time) const {
        time = ClampTime(time);
        const node_v& spline = Select(type, coord);
        for (int i = 0; i < (int)times_.size(); i++)
            if (time < times_.at(i).time && spline.at(i).type != Empty) return i;
        if (time == times_.back().time) return (int)times_.size() - 1;
        throw std::runtime_error("Invalid time.");
    }

    // :1114-1128
    int ConvertContactNodeToSplineNode(int contact_idx) const {
        int contacts = 0;
        for (int i = 0; i < (int)times_.size(); i++) {
            if (contacts == contact_idx && times_.at(i).type != Inter) return i;
            if (times_.at(i).type == LiftOff || times_.at(i).type == TouchDown) contacts++;
        }
        throw std::runtime_error("not a valid contact index.");
    }

    const time_v& RawTimes() const { return times_; }
    const node_v& RawNodes(SplineType type, int coord) const { return Select(type, coord); }

private:
    void push_pattern(NodeType f, NodeType p, NodeType z, TimeType t) {
        force_pat_.push_back(f); pos_pat_.push_back(p); zpos_pat_.push_back(z); time_pat_.push_back(t);
    }
    double ClampTime(double time) const {
        if (time < times_.front().time && time - times_.front().time >= -1e-4) time = times_.front().time;
        else if (time < times_.front().time) throw std::runtime_error("Time requested is too small.");
        if (time > times_.back().time && time - times_.back().time <= 1e4) time = times_.back().time;
        else if (time > times_.back().time) throw std::runtime_error("Time requested is too large.");
        return time;
    }
    int CountFullDerivBack(int coord, int lower_node) const {   // the `while (force_node == FullDeriv)` walk, :602-609
        int j = 0, idx = lower_node;
        while (forces_[coord].at(idx).type == FullDeriv) { j++; idx--; }
        return j;
    }
    node_v& Select(SplineType type, int coord) { return type == Force ? forces_[coord] : positions_[coord]; }
    const node_v& Select(SplineType type, int coord) const { return type == Force ? forces_[coord] : positions_[coord]; }

    // Hermite basis, :1179-1197
    static double x0Coef(double t, double dt) { return 1 - (1 / std::pow(dt, 2)) * 3 * std::pow(t, 2) + (1 / std::pow(dt, 3)) * 2 * std::pow(t, 3); }
    static double x1Coef(double t, double dt) { return (1 / std::pow(dt, 2)) * 3 * std::pow(t, 2) - (1 / std::pow(dt, 3)) * 2 * std::pow(t, 3); }
    static double x0dotCoef(double t, double dt) { return t - (1 / dt) * 2 * std::pow(t, 2) + (1 / std::pow(dt, 2)) * std::pow(t, 3); }
    static double x1dotCoef(double t, double dt) { return -(1 / dt) * std::pow(t, 2) + (1 / std::pow(dt, 2)) * std::pow(t, 3); }
    // basis partials, :1199-1244
    static double x0CoefPartial(double t, double D, double dtdth, double dD) {
        return (6 * std::pow(D, -3) * std::pow(t, 2) - 6 * std::pow(D, -4) * std::pow(t, 3)) * dD +
               (-6 * std::pow(D, -2) * t + 6 * std::pow(D, -3) * std::pow(t, 2)) * dtdth;
    }
    static double x1CoefPartial(double t, double D, double dtdth, double dD) {
        return (-6 * std::pow(D, -3) * std::pow(t, 2) + 6 * std::pow(D, -4) * std::pow(t, 3)) * dD +
               (6 * std::pow(D, -2) * t - 6 * std::pow(D, -3) * std::pow(t, 2)) * dtdth;
    }
    static double x0dotCoefPartial(double t, double D, double dtdth, double dD) {
        return (2 * std::pow(D, -2) * std::pow(t, 2) - 2 * std::pow(D, -3) * std::pow(t, 3)) * dD +
               (1 - std::pow(D, -1) * 4 * t + std::pow(D, -2) * 3 * std::pow(t, 2)) * dtdth;
    }
    static double x1dotCoefPartial(double t, double D, double dtdth, double dD) {
        return (std::pow(D, -2) * std::pow(t, 2) - 2 * std::pow(D, -3) * std::pow(t, 3)) * dD +
               (-std::pow(D, -1) * 2 * t + std::pow(D, -2) * 3 * std::pow(t, 2)) * dtdth;
    }

    std::array<node_v, 3> forces_, positions_;
    time_v times_;
    int num_force_polys_, spline_stride_;
    std::vector<NodeType> force_pat_, pos_pat_, zpos_pat_;
    std::vector<TimeType> time_pat_;
};

}  // namespace orc
