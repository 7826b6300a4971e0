'use strict';
// Config of the FlexLight API (reference modules/config.js:3-16), same field names and defaults; the
// renderer re-reads it every frame.  The HIP renderer has no history pass and no post-AA yet, so
// `temporal` and `antialiasing` default to off here (the reference's defaults are true / 'fxaa').
class Config {
  constructor () {
    this.samplesPerRay = 1;
    this.renderQuality = 1;
    this.maxReflections = 5;
    this.minImportancy = 0.3;
    this.firstPasses = 3;
    this.secondPasses = 3;
    this.temporal = false;
    this.temporalSamples = 4;
    this.filter = false;
    this.hdr = true;
    this.antialiasing = undefined;
  }
}
module.exports = { Config };
