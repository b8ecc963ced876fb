// Same relative path as the reference header (src/models/tokenizer.h); the implementation lives in api/tokenizer.hpp
#pragma once
#include "../../api/tokenizer.hpp"
