// Same relative path as the reference header; the implementation lives in api/layers.hpp
#pragma once
#include "../../../api/layers.hpp"
