#pragma once
#include <google/protobuf/message.h>
