// Same relative path as the reference header; the implementation lives in api/model.hpp
#pragma once
#include "../../api/model.hpp"
