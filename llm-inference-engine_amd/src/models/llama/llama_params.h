// Same relative path as the reference header; the implementation lives in api/params.hpp
#pragma once
#include "../../../api/params.hpp"
