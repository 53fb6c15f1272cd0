// host_tree.hpp - the object behind lvbhost_tree (include/lvbhost.h)
#pragma once

#include <vector>

#include "program.hpp"
#include "proposals.hpp"

struct lvbhost_tree
{
    lvbgpu::Topology topo;
    lvbgpu::Rng rng;
    lvbgpu::ProgramBuilder pb;
    std::vector<lvbgpu::Edit> scratch;
};
