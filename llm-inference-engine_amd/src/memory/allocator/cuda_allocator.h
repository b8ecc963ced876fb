// Same relative path as the reference header; the implementation lives in api/allocator.hpp
#pragma once
#include "../../../api/allocator.hpp"
