// Source-compatible include path: code written against libmpc++ says `#include <mpc/LMPC.hpp>`.
// The implementation lives in include/mpcx/LMPC.hpp (mpc::LMPC<> over the mpcx C ABI).
#pragma once
#include "../mpcx/LMPC.hpp"
