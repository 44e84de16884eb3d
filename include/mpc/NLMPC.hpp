// drop-in include path: the reference header is <mpc/NLMPC.hpp>
#pragma once
#include "../mpcx/NLMPC.hpp"
