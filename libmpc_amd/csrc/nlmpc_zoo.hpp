// The model zoo of mpcx/nlmpc_models.hpp by id (mpcx_nlmpc_create): shared by the translation units that instantiate the engine for it --
// nlmpc_kernels.hip (transcription, nlmpc_sqp) and nlmpc_wg_kernels.hip (nlmpc_sqp_wg); two files so that the two kernel families compile side by side.
#pragma once
#include "mpcx/nlmpc_models.hpp"

namespace mpcx {
namespace {

using namespace models;

template <class F>
int dispatch_model(int model_id, F &&fn)
{
    switch (model_id) {
    case 1: return fn(VanDerPol{});
    case 2: return fn(Ugv{});
    case 3: return fn(Oscillators<6>{});
    case 4: return fn(Oscillators<8>{});
    case 5: return fn(VanDerPolTerminal{});
    case 6: return fn(VanDerPolRate{});
    default: return -1;
    }
}

}  // namespace
}  // namespace mpcx
