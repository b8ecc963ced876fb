// Same relative path as the reference header; the implementation lives in api/tensor.hpp
#pragma once
#include "../../api/tensor.hpp"
