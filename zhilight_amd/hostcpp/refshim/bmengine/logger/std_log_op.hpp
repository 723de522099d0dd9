// refshim: stream insertion for the std containers the reference's host code prints in its assert / log messages
// (3rd/bmengine/bmengine/include/bmengine/logger/std_log_op.hpp: vectors as [a, b, c], pairs / tuples as (a, b)).
#pragma once
#include <ostream>
#include <sstream>
#include <string>
#include <tuple>
#include <utility>
#include <vector>

namespace bmengine {
namespace logger {

template <class T> inline std::ostream& operator<<(std::ostream& out, const std::vector<T>& v) {
    out << "[";
    for (size_t i = 0; i < v.size(); ++i) out << (i ? ", " : "") << v[i];
    return out << "]";
}
template <class A, class B> inline std::ostream& operator<<(std::ostream& out, const std::pair<A, B>& p) {
    return out << "(" << p.first << ", " << p.second << ")";
}
template <class... Ts> inline std::ostream& operator<<(std::ostream& out, const std::tuple<Ts...>& t) {
    out << "(";
    std::apply([&](const Ts&... xs) { size_t i = 0; ((out << (i++ ? ", " : "") << xs), ...); }, t);
    return out << ")";
}

template <class T> inline std::string to_string(const std::vector<T>& v) {
    std::ostringstream os;
    os << v;
    return os.str();
}
using std::to_string;
static inline std::string to_string(const char* s) { return s; }
static inline std::string to_string(const std::string& s) { return s; }
template <class A> inline std::string str_cat(const A& a) { return to_string(a); }
template <class A, class B, class... Rest> inline std::string str_cat(const A& a, const B& b, const Rest&... rest) {
    return to_string(a) + str_cat(b, rest...);
}

}  // namespace logger
}  // namespace bmengine

namespace std {
using bmengine::logger::operator<<;
using bmengine::logger::to_string;
}
using std::endl;
