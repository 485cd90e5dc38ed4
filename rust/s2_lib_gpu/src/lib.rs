//! `s2_lib::try3::synth`-shaped wrapper over the C ABI of libs2r (include/s2r.h).
//!
//! UNVERIFIED: written without a Rust toolchain (none in the build image); see INTEGRATION.md.
//!
//! The public surface is exactly what `s2_bin` uses today
//! (components/s2_bin/src/main.rs:13,125,132-133,142,147,191,199-205):
//!
//! ```ignore
//! use s2_lib_gpu::synth::{Synth, Note, Velocity};
//! use s2_lib_gpu::units::{SampleRateKhz, Unipolar};
//! let mut synth = Synth::new();
//! synth.note_on(Note(69), Velocity(Unipolar(1.0)));
//! synth.sample(&mut chunk, SampleRateKhz(48_000));
//! synth.note_off(Note(69));
//! ```

pub mod units {
    /// components/s2_lib/src/try3/units.rs:11
    #[derive(Copy, Clone)]
    pub struct Unipolar<const N: u16>(pub f32);
    /// units.rs:14 — holds Hz despite the name (units.rs:21)
    #[derive(Copy, Clone)]
    pub struct SampleRateKhz(pub u32);
}

pub mod ffi {
    use std::os::raw::{c_char, c_int, c_void};

    #[repr(C)]
    pub struct S2rConfig {
        pub struct_size: u32,
        pub total_voices: u32,
        pub shard_begin: u32,
        pub shard_voices: u32,
        pub max_frames: u32,
        pub device: i32,
        pub block_voices: u32,
        pub mix_groups: u32,
        pub lanes_per_voice: u32,
        pub shard_interleave: u32,
        pub shard_index: u32,
        pub shard_count: u32,
    }

    #[repr(C)]
    pub struct S2rSynth {
        _private: [u8; 0],
    }

    /// include/s2r.h `s2r_note_event`: `frame` = 0 or the 16-frame boundary inside the next fill
    #[repr(C)]
    #[derive(Copy, Clone)]
    pub struct S2rNoteEvent {
        pub kind: u8, // 0 = note_off, 1 = note_on
        pub note: u8,
        pub frame: u16,
        pub velocity: f32,
    }

    extern "C" {
        pub fn s2r_note_events(s: *mut S2rSynth, events: *const S2rNoteEvent, n: usize) -> c_int;
        pub fn s2r_create(cfg: *const S2rConfig, out: *mut *mut S2rSynth) -> c_int;
        pub fn s2r_destroy(s: *mut S2rSynth);
        pub fn s2r_load_patch(s: *mut S2rSynth, text: *const c_char, len: usize) -> c_int;
        pub fn s2r_program_change(s: *mut S2rSynth, program: u32) -> c_int;
        pub fn s2r_note_on(s: *mut S2rSynth, note: u8, velocity: f32) -> c_int;
        pub fn s2r_note_off(s: *mut S2rSynth, note: u8) -> c_int;
        pub fn s2r_fill(s: *mut S2rSynth, mono_out: *mut f32, frames: usize, sample_rate_hz: u32) -> c_int;
        pub fn s2r_fill_device(s: *mut S2rSynth, dev_out: *mut f32, frames: usize, sample_rate_hz: u32,
                               hip_stream: *mut c_void) -> c_int;
        pub fn s2r_last_error(s: *const S2rSynth) -> *const c_char;
        pub fn s2r_status_string(status: c_int) -> *const c_char;
    }
}

pub mod synth {
    use super::ffi;
    use super::units::{SampleRateKhz, Unipolar};
    use std::ffi::CStr;

    /// components/s2_lib/src/try3/synth.rs:7 — the reference's fixed pool size
    pub const NUM_VOICES: u32 = 8;
    /// largest slice `sample` is ever handed by s2_bin (audio_player.rs:21 uses 2048-frame buffers)
    pub const MAX_FRAMES: u32 = 2048;

    /// synth.rs:16
    #[derive(Eq, PartialEq, Copy, Clone)]
    pub struct Note(pub u8);
    /// synth.rs:18
    #[derive(Copy, Clone)]
    pub struct Velocity(pub Unipolar<1>);

    pub struct Synth {
        handle: *mut ffi::S2rSynth,
    }

    // like the reference's Synth (plain data behind &mut), the handle may move between threads
    // but must only be used by one at a time
    unsafe impl Send for Synth {}

    fn fail(handle: *const ffi::S2rSynth, status: i32) -> ! {
        let msg = unsafe {
            if handle.is_null() {
                CStr::from_ptr(ffi::s2r_status_string(status))
            } else {
                CStr::from_ptr(ffi::s2r_last_error(handle))
            }
        };
        // the reference panics in the same situations (process.rs:36,71 `expect("overflow")`)
        panic!("libs2r status {}: {}", status, msg.to_string_lossy());
    }

    impl Synth {
        /// synth.rs:54-59
        pub fn new() -> Synth {
            Synth::with_voices(NUM_VOICES)
        }

        pub fn with_voices(voices: u32) -> Synth {
            let cfg = ffi::S2rConfig {
                struct_size: std::mem::size_of::<ffi::S2rConfig>() as u32,
                total_voices: voices,
                shard_begin: 0,
                shard_voices: 0,
                max_frames: MAX_FRAMES,
                device: -1,
                block_voices: 0,
                mix_groups: 0,
                lanes_per_voice: 0,
                shard_interleave: 0,
                shard_index: 0,
                shard_count: 1,
            };
            let mut handle = std::ptr::null_mut();
            let rc = unsafe { ffi::s2r_create(&cfg, &mut handle) };
            if rc != 0 {
                fail(std::ptr::null(), rc);
            }
            Synth { handle }
        }

        /// Multi-timbral extension: the patch (bank index) the following note_ons use.
        pub fn program_change(&mut self, program: u32) {
            let rc = unsafe { ffi::s2r_program_change(self.handle, program) };
            if rc != 0 {
                fail(self.handle, rc);
            }
        }

        /// example.synth2 text (the reference has no loader; an empty body is default_config())
        pub fn load_patch(&mut self, text: &str) {
            let rc = unsafe { ffi::s2r_load_patch(self.handle, text.as_ptr() as *const _, text.len()) };
            if rc != 0 {
                fail(self.handle, rc);
            }
        }

        /// synth.rs:61-70
        pub fn note_on(&mut self, note: Note, velocity: Velocity) {
            let rc = unsafe { ffi::s2r_note_on(self.handle, note.0, (velocity.0).0) };
            if rc != 0 {
                fail(self.handle, rc);
            }
        }

        /// synth.rs:72-80
        pub fn note_off(&mut self, note: Note) {
            let rc = unsafe { ffi::s2r_note_off(self.handle, note.0) };
            if rc != 0 {
                fail(self.handle, rc);
            }
        }

        /// Events stamped with the 16-frame boundary (`frame`) at which s2_bin's loop
        /// (main.rs:138-143) would have applied them; they take effect inside the next `sample`.
        pub fn note_events(&mut self, events: &[ffi::S2rNoteEvent]) {
            let rc = unsafe { ffi::s2r_note_events(self.handle, events.as_ptr(), events.len()) };
            if rc != 0 {
                fail(self.handle, rc);
            }
        }

        /// synth.rs:154-169 — overwrites `buffer`
        pub fn sample(&mut self, buffer: &mut [f32], sample_rate: SampleRateKhz) {
            let rc = unsafe { ffi::s2r_fill(self.handle, buffer.as_mut_ptr(), buffer.len(), sample_rate.0) };
            if rc != 0 {
                fail(self.handle, rc);
            }
        }
    }

    impl Drop for Synth {
        fn drop(&mut self) {
            unsafe { ffi::s2r_destroy(self.handle) }
        }
    }
}
