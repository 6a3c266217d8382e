// stand-in (declarations only): see tests/refhdr/spdlog/spdlog.h
#pragma once
