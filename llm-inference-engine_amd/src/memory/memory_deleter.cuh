// Same relative path as the reference header; the implementation lives in api/runtime.hpp
#pragma once
#include "../api/runtime.hpp"
