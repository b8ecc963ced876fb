// Same relative path as the reference header; the implementation lives in api/weights.hpp
#pragma once
#include "../../../api/weights.hpp"
