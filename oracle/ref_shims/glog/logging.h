// DEPENDENCY SHIM (oracle/_ref build only): the subset of glog that the voxblox hot-path
// sources use.  CHECK failures and LOG(FATAL) abort, like glog.  Written for this repo;
// not glog code.
#pragma once
#include <cmath>
#include <cstdlib>
#include <cstring>  // real glog pulls this in; block.cc relies on it for memcpy
#include <iostream>
#include <sstream>

namespace vbxshim {
struct LogSink {
  bool fatal;
  std::ostringstream os;
  explicit LogSink(bool f) : fatal(f) {}
  ~LogSink() {
    if (fatal) {
      std::cerr << os.str() << std::endl;
      std::abort();
    }
  }
  template <typename T> LogSink& operator<<(const T& v) { os << v; return *this; }
  LogSink& operator<<(std::ostream& (*f)(std::ostream&)) { os << f; return *this; }
};
struct Voidify { void operator&(const LogSink&) {} };
template <typename T> T& check_notnull(T& p, const char* what) {
  if (p == nullptr) { std::cerr << "Check failed: '" << what << "' must be non NULL" << std::endl; std::abort(); }
  return p;
}
template <typename T> T* check_notnull(T* p, const char* what) {
  if (p == nullptr) { std::cerr << "Check failed: '" << what << "' must be non NULL" << std::endl; std::abort(); }
  return p;
}
}  // namespace vbxshim

#define VBXSHIM_LOG_IF(fatal, cond) !(cond) ? (void)0 : ::vbxshim::Voidify() & ::vbxshim::LogSink(fatal)
#define LOG(severity) VBXSHIM_LOG_##severity
#define VBXSHIM_LOG_INFO VBXSHIM_LOG_IF(false, false)
#define VBXSHIM_LOG_WARNING VBXSHIM_LOG_IF(false, false)
#define VBXSHIM_LOG_ERROR VBXSHIM_LOG_IF(false, false)
#define VBXSHIM_LOG_FATAL VBXSHIM_LOG_IF(true, true)
#define VLOG(n) VBXSHIM_LOG_IF(false, false)
#define LOG_FIRST_N(severity, n) VBXSHIM_LOG_IF(false, false)
#define LOG_EVERY_N(severity, n) VBXSHIM_LOG_IF(false, false)
#define CHECK(cond) VBXSHIM_LOG_IF(true, !(cond)) << "Check failed: " #cond " "
#define CHECK_OP(a, b, op) VBXSHIM_LOG_IF(true, !((a)op(b))) << "Check failed: " #a " " #op " " #b " "
#define CHECK_EQ(a, b) CHECK_OP(a, b, ==)
#define CHECK_NE(a, b) CHECK_OP(a, b, !=)
#define CHECK_LT(a, b) CHECK_OP(a, b, <)
#define CHECK_LE(a, b) CHECK_OP(a, b, <=)
#define CHECK_GT(a, b) CHECK_OP(a, b, >)
#define CHECK_GE(a, b) CHECK_OP(a, b, >=)
#define CHECK_NEAR(a, b, tol) VBXSHIM_LOG_IF(true, !(std::abs((a) - (b)) <= (tol))) << "Check failed: near "
#define CHECK_NOTNULL(p) ::vbxshim::check_notnull((p), #p)
// Release build: DCHECKs compile out (the reference's catkin Release build defines NDEBUG).
#define DCHECK(cond) VBXSHIM_LOG_IF(false, false)
#define DCHECK_EQ(a, b) VBXSHIM_LOG_IF(false, false)
#define DCHECK_NE(a, b) VBXSHIM_LOG_IF(false, false)
#define DCHECK_LT(a, b) VBXSHIM_LOG_IF(false, false)
#define DCHECK_LE(a, b) VBXSHIM_LOG_IF(false, false)
#define DCHECK_GT(a, b) VBXSHIM_LOG_IF(false, false)
#define DCHECK_GE(a, b) VBXSHIM_LOG_IF(false, false)
#define DCHECK_NOTNULL(p) (p)
