#pragma once
#include "../../include/maai_hip.h"
