// Same relative path as the reference header; the implementation lives in api/kernels.hpp
#pragma once
#include "../../../api/kernels.hpp"
