// Build-only stand-in for the reference's <network/mqtt.h> (which needs the Paho MQTT C++ client, an un-vendored apt
// dependency, Dockerfile:4). TEST INFRASTRUCTURE: lets oracle/Makefile compile the reference's own
// network/data_controller.cpp and radio/blocks/spectrogram.cpp in place. It declares the one class those sources use;
// publish() keeps what would have gone to the broker so that ref_shim.cpp can hand the bytes to the tests.
#pragma once
#include <config.h>
#include <logger.h>

#include <cstdint>
#include <deque>
#include <string>
#include <utility>
#include <vector>

class Mqtt {
 public:
  Mqtt() {}
  void publish(const std::string& topic, const std::string& data, int = 0) { m_sent.emplace_back(topic, std::vector<uint8_t>(data.begin(), data.end())); }
  void publish(const std::string& topic, const std::vector<uint8_t>& data, int = 0) { m_sent.emplace_back(topic, data); }
  void publish(const std::string& topic, const std::vector<uint8_t>&& data, int = 0) { m_sent.emplace_back(topic, data); }

  std::deque<std::pair<std::string, std::vector<uint8_t>>> m_sent;
};
