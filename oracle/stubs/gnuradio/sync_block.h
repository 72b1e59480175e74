// Build-only stand-in for <gnuradio/sync_block.h> (GNU Radio is an un-vendored apt dependency of the
// reference, Dockerfile:4). TEST INFRASTRUCTURE: lets oracle/Makefile compile the reference's own
// block sources in place, from /root/reference/sources, without GNU Radio installed. It declares only
// the names those sources use: gr_complex, the two item-vector typedefs, io_signature::make and an
// abstract sync_block with a virtual work().
#pragma once
#include <complex>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

typedef std::complex<float> gr_complex;
typedef std::vector<const void*> gr_vector_const_void_star;
typedef std::vector<void*> gr_vector_void_star;

namespace gr {
class io_signature {
 public:
  typedef std::shared_ptr<io_signature> sptr;
  static sptr make(int min_streams, int max_streams, int sizeof_stream_item) {
    auto s = std::make_shared<io_signature>();
    s->m_min = min_streams;
    s->m_max = max_streams;
    s->m_item = sizeof_stream_item;
    return s;
  }
  int sizeof_stream_item(int) const { return m_item; }
  int m_min = 0, m_max = 0, m_item = 0;
};

class sync_block {
 public:
  sync_block() {}
  sync_block(const std::string& name, io_signature::sptr in, io_signature::sptr out) : m_name(name), m_in(in), m_out(out) {}
  virtual ~sync_block() {}
  virtual int work(int noutput_items, gr_vector_const_void_star& input_items, gr_vector_void_star& output_items) = 0;
  const std::string& name() const { return m_name; }
  io_signature::sptr input_signature() const { return m_in; }
  io_signature::sptr output_signature() const { return m_out; }

 private:
  std::string m_name;
  io_signature::sptr m_in, m_out;
};
}  // namespace gr
