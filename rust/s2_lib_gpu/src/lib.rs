//! `s2_lib::try3::synth`-shaped wrapper over the C ABI of libs2r (include/s2r.h, S2R_ABI_VERSION 4).
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
//!
//! plus what the boundary adds: a run-time pool size (`with_voices`), several GPUs behind one `Synth`
//! (`with_devices`), events stamped with their 16-frame boundary (`note_events`), the fill in two halves for a caller
//! with two buffers in flight (`sample_begin` / `sample_end`), patch text and patch banks, the stereo copy and the
//! 4x-oversampled fill.

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

    /// include/s2r.h `S2R_ABI_VERSION`
    pub const S2R_ABI_VERSION: u32 = 4;
    /// include/s2r.h `S2R_MAX_DEVICES`
    pub const S2R_MAX_DEVICES: usize = 16;

    /// include/s2r.h `s2r_config`
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
        pub reserved0: u32,
        pub shard_interleave: u32,
        pub shard_index: u32,
        pub shard_count: u32,
        pub n_devices: u32,
        pub devices: [i32; S2R_MAX_DEVICES],
    }

    /// include/s2r.h `s2r_adsr` (static_config.rs:38-44)
    #[repr(C)]
    #[derive(Copy, Clone)]
    pub struct S2rAdsr {
        pub attack_ms: f32,
        pub decay_ms: f32,
        pub sustain: f32,
        pub release_ms: f32,
    }

    /// include/s2r.h `s2r_patch` (static_config.rs:4-24 plus the build-defined filter selection)
    #[repr(C)]
    #[derive(Copy, Clone)]
    pub struct S2rPatch {
        pub osc_kind: i32,
        pub osc_gain: f32,
        pub noise: f32,
        pub lpf_freq: f32,
        pub amp_env: S2rAdsr,
        pub mod_env: S2rAdsr,
        pub mod_env_to_osc_freq: f32,
        pub mod_env_to_lpf_freq: f32,
        pub lpf_kind: i32,
        pub lpf_damping: f32,
        pub lpf_q: f32,
    }

    #[repr(C)]
    pub struct S2rSynth {
        _private: [u8; 0],
    }

    /// include/s2r.h `s2r_layer_call`: one call of process::process_layer_buf_simd (process.rs:14-22), st::Layer in place
    #[repr(C)]
    #[derive(Copy, Clone)]
    pub struct S2rLayerCall {
        pub pitch_hz: f32,
        pub offset: u32,
        pub release_offset: u32,
        pub has_release: u8,
        pub program: u8,
        pub _pad: [u8; 2],
        pub phase_accum: f32,
        pub lpf_last: f32,
        pub noise_seed: u32,
        pub filt_x1: f32,
        pub filt_x2: f32,
        pub filt_y1: f32,
        pub filt_y2: f32,
        pub osc_z: f32,
    }

    /// include/s2r.h `s2r_note_event`: `frame` = 0 or the 16-frame boundary inside the next fill
    #[repr(C)]
    #[derive(Copy, Clone)]
    pub struct S2rNoteEvent {
        pub kind: u8, // 0 = note_off, 1 = note_on, 2 = program change (`note` = bank index)
        pub note: u8,
        pub frame: u16,
        pub velocity: f32,
    }

    extern "C" {
        pub fn s2r_abi_version() -> u32;
        pub fn s2r_create(cfg: *const S2rConfig, out: *mut *mut S2rSynth) -> c_int;
        pub fn s2r_destroy(s: *mut S2rSynth);
        pub fn s2r_load_patch(s: *mut S2rSynth, text: *const c_char, len: usize) -> c_int;
        pub fn s2r_set_patch(s: *mut S2rSynth, patch: *const S2rPatch) -> c_int;
        pub fn s2r_get_patch(s: *const S2rSynth, out: *mut S2rPatch) -> c_int;
        pub fn s2r_default_patch(out: *mut S2rPatch);
        pub fn s2r_set_patch_bank(s: *mut S2rSynth, patches: *const S2rPatch, n: u32) -> c_int;
        pub fn s2r_program_change(s: *mut S2rSynth, program: u32) -> c_int;
        pub fn s2r_note_on(s: *mut S2rSynth, note: u8, velocity: f32) -> c_int;
        pub fn s2r_note_off(s: *mut S2rSynth, note: u8) -> c_int;
        pub fn s2r_note_events(s: *mut S2rSynth, events: *const S2rNoteEvent, n: usize) -> c_int;
        pub fn s2r_fill(s: *mut S2rSynth, mono_out: *mut f32, frames: usize, sample_rate_hz: u32) -> c_int;
        pub fn s2r_fill_begin(s: *mut S2rSynth, frames: usize, sample_rate_hz: u32) -> c_int;
        pub fn s2r_fill_end(s: *mut S2rSynth, mono_out: *mut f32, capacity: usize) -> c_int;
        pub fn s2r_fill_pending_frames(s: *const S2rSynth) -> usize;
        pub fn s2r_fills_in_flight(s: *const S2rSynth) -> u32;
        pub fn s2r_fill_stereo(s: *mut S2rSynth, interleaved_lr_out: *mut f32, frames: usize, sample_rate_hz: u32) -> c_int;
        pub fn s2r_fill_oversampled(s: *mut S2rSynth, mono_out: *mut f32, frames: usize, sample_rate_hz: u32) -> c_int;
        pub fn s2r_fill_device(s: *mut S2rSynth, dev_out: *mut f32, frames: usize, sample_rate_hz: u32,
                               hip_stream: *mut c_void) -> c_int;
        pub fn s2r_device_count(s: *const S2rSynth) -> u32;
        pub fn s2r_set_low_latency(s: *mut S2rSynth, enabled: c_int) -> c_int;
        pub fn s2r_set_resident(s: *mut S2rSynth, enabled: c_int) -> c_int;
        pub fn s2r_resident_active(s: *const S2rSynth) -> c_int;
        pub fn s2r_quiesce(s: *mut S2rSynth) -> c_int;
        pub fn s2r_exchange_create(s: *mut S2rSynth, n_ranks: u32, handle_out: *mut c_void, handle_bytes: usize) -> c_int;
        pub fn s2r_exchange_attach(s: *mut S2rSynth, rank: u32, n_ranks: u32, handle: *const c_void, handle_bytes: usize) -> c_int;
        pub fn s2r_process_layers(s: *mut S2rSynth, layers: *mut S2rLayerCall, n_layers: u32, bufs: *mut f32, frames: usize, sample_rate_hz: u32) -> c_int;
        pub fn s2r_set_voice_log(s: *mut S2rSynth, f: Option<extern "C" fn(*mut c_void, u32, u8)>, user: *mut c_void) -> c_int;
        pub fn s2r_build_id() -> *const c_char;
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
            Synth::create(voices, &[], 0)
        }

        /// One `Synth` over several GPUs: the pool is dealt out to `devices` in runs of 64 voices, the allocation
        /// policy (synth.rs:61-120) runs once per event on the calling thread, and the shards' partial mixes are
        /// added in shard order on `devices[0]`.  `voices` must be a multiple of 256 x the number of devices.
        pub fn with_devices(voices: u32, devices: &[i32]) -> Synth {
            Synth::create(voices, devices, 64)
        }

        fn create(voices: u32, devices: &[i32], interleave: u32) -> Synth {
            if unsafe { ffi::s2r_abi_version() } != ffi::S2R_ABI_VERSION {
                panic!("libs2r: ABI version mismatch (this shim was written for {})", ffi::S2R_ABI_VERSION);
            }
            assert!(devices.len() <= ffi::S2R_MAX_DEVICES);
            let mut list = [0i32; ffi::S2R_MAX_DEVICES];
            list[..devices.len()].copy_from_slice(devices);
            let cfg = ffi::S2rConfig {
                struct_size: std::mem::size_of::<ffi::S2rConfig>() as u32,
                total_voices: voices,
                shard_begin: 0,
                shard_voices: 0,
                max_frames: MAX_FRAMES,
                device: -1,
                block_voices: 0,
                mix_groups: 0,
                reserved0: 0,
                shard_interleave: if devices.len() > 1 { interleave } else { 0 },
                shard_index: 0,
                shard_count: 1,
                n_devices: devices.len() as u32,
                devices: list,
            };
            let mut handle = std::ptr::null_mut();
            let rc = unsafe { ffi::s2r_create(&cfg, &mut handle) };
            if rc != 0 {
                fail(std::ptr::null(), rc);
            }
            Synth { handle }
        }

        fn check(&self, rc: i32) {
            if rc != 0 {
                fail(self.handle, rc);
            }
        }

        /// Multi-timbral extension: the patch (bank index) the following note_ons use.
        pub fn program_change(&mut self, program: u32) {
            self.check(unsafe { ffi::s2r_program_change(self.handle, program) });
        }

        /// example.synth2 text (the reference has no loader; an empty body is default_config())
        pub fn load_patch(&mut self, text: &str) {
            self.check(unsafe { ffi::s2r_load_patch(self.handle, text.as_ptr() as *const _, text.len()) });
        }

        /// static_config::Layer as a value (synth.rs:10 `config`)
        pub fn set_patch(&mut self, patch: &ffi::S2rPatch) {
            self.check(unsafe { ffi::s2r_set_patch(self.handle, patch) });
        }

        /// A bank of 1..=256 patches; `program_change` picks the one a note_on gives its voice.
        pub fn set_patch_bank(&mut self, patches: &[ffi::S2rPatch]) {
            self.check(unsafe { ffi::s2r_set_patch_bank(self.handle, patches.as_ptr(), patches.len() as u32) });
        }

        /// synth.rs:61-70
        pub fn note_on(&mut self, note: Note, velocity: Velocity) {
            self.check(unsafe { ffi::s2r_note_on(self.handle, note.0, (velocity.0).0) });
        }

        /// synth.rs:72-80
        pub fn note_off(&mut self, note: Note) {
            self.check(unsafe { ffi::s2r_note_off(self.handle, note.0) });
        }

        /// Events stamped with the 16-frame boundary (`frame`) at which s2_bin's loop
        /// (main.rs:138-143) would have applied them; they take effect inside the next `sample`.
        pub fn note_events(&mut self, events: &[ffi::S2rNoteEvent]) {
            self.check(unsafe { ffi::s2r_note_events(self.handle, events.as_ptr(), events.len()) });
        }

        /// synth.rs:154-169 — overwrites `buffer`
        pub fn sample(&mut self, buffer: &mut [f32], sample_rate: SampleRateKhz) {
            self.check(unsafe { ffi::s2r_fill(self.handle, buffer.as_mut_ptr(), buffer.len(), sample_rate.0) });
        }

        /// First half of `sample` for a caller that keeps two buffers in flight, as s2_bin does between its synth and
        /// audio threads (audio_player.rs:56-60, main.rs:135-149): queues a fill of `frames`; at most two may be
        /// queued.
        pub fn sample_begin(&mut self, frames: usize, sample_rate: SampleRateKhz) {
            self.check(unsafe { ffi::s2r_fill_begin(self.handle, frames, sample_rate.0) });
        }

        /// Second half: waits for the OLDEST queued fill and overwrites the front of `buffer` with its frames;
        /// returns how many.  Panics (like a slice length mismatch in the reference would) when `buffer` is shorter
        /// than that fill.
        pub fn sample_end(&mut self, buffer: &mut [f32]) -> usize {
            let frames = unsafe { ffi::s2r_fill_pending_frames(self.handle) };
            self.check(unsafe { ffi::s2r_fill_end(self.handle, buffer.as_mut_ptr(), buffer.len()) });
            frames
        }

        /// The audio callback's mono -> every channel copy (audio_player.rs:224-228) done on the device:
        /// `interleaved` holds L, R pairs with L == R; its length is twice the frame count.
        pub fn sample_stereo(&mut self, interleaved: &mut [f32], sample_rate: SampleRateKhz) {
            assert!(interleaved.len() % 2 == 0);
            self.check(unsafe {
                ffi::s2r_fill_stereo(self.handle, interleaved.as_mut_ptr(), interleaved.len() / 2, sample_rate.0)
            });
        }

        /// Build-defined: rendered at 4 x `sample_rate` and decimated to `buffer.len()` frames.
        pub fn sample_oversampled(&mut self, buffer: &mut [f32], sample_rate: SampleRateKhz) {
            self.check(unsafe {
                ffi::s2r_fill_oversampled(self.handle, buffer.as_mut_ptr(), buffer.len(), sample_rate.0)
            });
        }

        /// For the reference's own call pattern — `sample()` per 16 frames from the audio callback (main.rs:138-147): keeps a
        /// resident render kernel on the device between calls, so that a fill is a command in mapped host memory, not a
        /// launch.  Small pools only (one workgroup: at most 256 voices); same samples either way.
        pub fn set_low_latency(&mut self, enabled: bool) {
            self.check(unsafe { ffi::s2r_set_low_latency(self.handle, if enabled { 1 } else { 0 }) });
        }

        /// The throughput caller's form of the same idea (`s2r_set_resident`, include/s2r.h): the shard's whole render grid
        /// stays on the device between fills — a fill is a posted command, not a launch — for `sample`, `sample_stereo` and
        /// `sample_begin` / `sample_end`; every shard of a `with_devices` Synth gets one.  Same samples either way.
        pub fn set_resident(&mut self, enabled: bool) {
            self.check(unsafe { ffi::s2r_set_resident(self.handle, if enabled { 1 } else { 0 }) });
        }

        /// stops any resident kernel of this Synth and waits for it (before the caller synchronises the whole device)
        pub fn quiesce(&mut self) {
            self.check(unsafe { ffi::s2r_quiesce(self.handle) });
        }

        /// One process per GPU without a collective in the step: rank 0 creates the exchange (the ranks' partial rows meet in
        /// a block of its device memory and its last workgroup adds them in rank order) and hands the 64 handle bytes to the
        /// other ranks, which `exchange_attach`.  Every rank then drives its shard's Synth with the same events and the same
        /// `sample` / `sample_begin` / `sample_end` calls; rank 0's buffers receive the mix.
        pub fn exchange_create(&mut self, n_ranks: u32) -> [u8; 64] {
            let mut h = [0u8; 64];
            self.check(unsafe { ffi::s2r_exchange_create(self.handle, n_ranks, h.as_mut_ptr() as *mut _, h.len()) });
            h
        }

        pub fn exchange_attach(&mut self, rank: u32, n_ranks: u32, handle: &[u8; 64]) {
            self.check(unsafe { ffi::s2r_exchange_attach(self.handle, rank, n_ranks, handle.as_ptr() as *const _, handle.len()) });
        }

        /// which sources the loaded libs2r was built from (`s2r_build_id`)
        pub fn build_id() -> String {
            unsafe { CStr::from_ptr(ffi::s2r_build_id()) }.to_string_lossy().into_owned()
        }

        /// how many GPUs render this Synth
        pub fn device_count(&self) -> u32 {
            unsafe { ffi::s2r_device_count(self.handle) }
        }

        /// `log::debug!("using new voice index {} for note {}", …)` of `next_voice` (synth.rs:118) through the `log` facade's
        /// place: `f(voice_index, note)` per note_on.  `None` switches it off.
        pub fn set_voice_log(&mut self, f: Option<extern "C" fn(*mut std::os::raw::c_void, u32, u8)>) {
            self.check(unsafe { ffi::s2r_set_voice_log(self.handle, f, std::ptr::null_mut()) });
        }

        /// process::process_layer_buf_simd (process.rs:14-49) for layers the caller keeps itself, `layers.len()` of them side
        /// by side: this Synth is the workspace (its patch is the `sc::Layer`, its voices are replaced), `bufs` is
        /// `[layers.len()][frames]`, every `S2rLayerCall`'s state fields are advanced in place.
        pub fn process_layers(&mut self, layers: &mut [ffi::S2rLayerCall], bufs: &mut [f32], frames: usize, sample_rate: SampleRateKhz) {
            assert!(bufs.len() >= layers.len() * frames);
            self.check(unsafe { ffi::s2r_process_layers(self.handle, layers.as_mut_ptr(), layers.len() as u32, bufs.as_mut_ptr(), frames, sample_rate.0) });
        }
    }

    impl Drop for Synth {
        fn drop(&mut self) {
            unsafe { ffi::s2r_destroy(self.handle) }
        }
    }
}
